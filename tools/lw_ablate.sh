# tuning-build ablations of the conv kernels' epilogue (GPU box): KDCC_CONV_TUNE 64 = no epilogue, 128 = no output store,
# 256 = stores into a small L2-resident window
export KDCC_LIB=tuning KDCC_BENCH_BATCH=8 KDCC_CONV_DUO=2
ONLY="mod2 3x3 128,mod4 3x3 512"
for cus in 256 128; do for t in 64 0; do echo "== duo, workgroups = 2 x $cus, TUNE=$t"; KDCC_PERSIST_CUS=$cus KDCC_CONV_TUNE=$t python tools/bench_conv.py --only "$ONLY" --iters 5 | grep -v weighted; done; done
