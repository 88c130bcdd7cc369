# tuning-build ablations of the conv kernels' epilogue (GPU box): KDCC_CONV_TUNE 64 = no epilogue, 128 = no output store,
# 256 = stores into a small L2-resident window, 32768 = write-through (sc1) output stores, 98304 = non-temporal
export KDCC_LIB=tuning KDCC_BENCH_BATCH=8
ONLY="mod3 3x3 256,mod4 3x3 512,mod5 3x3 d2 512,1x1 1024->2048,1x1 256->4096,1x1 2048->4096,mod2 3x3 128"
for r in 1 2; do for t in 0 32768 98304; do echo "== TUNE=$t"; KDCC_CONV_TUNE=$t python tools/bench_conv.py --only "$ONLY" --iters 5 | grep -v weighted; done; done
