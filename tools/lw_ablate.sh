# tuning-build ablations of the conv kernels' epilogue (GPU box, from the repo root): KDCC_CONV_TUNE 64 = no epilogue, 128 = no output
# store, 256 = stores into a small L2-resident window.  Arms: the lone-wave kernels (3x3 default, 1x1 opt-in) / the ping-pong kernels;
# then conv_row_duo_kernel with two and with one workgroup per CU.
export KDCC_LIB=tuning KDCC_BENCH_BATCH=8
ONLY="mod2 3x3 128,mod3 3x3 256,mod4 3x3 512,1x1 1024->2048,1x1 256->4096"
for lw in 1 0; do for t in 0 64 128 256; do echo "== LW=$lw TUNE=$t"; KDCC_CONV_LW=$lw KDCC_CONV_LW_PW=$lw KDCC_CONV_TUNE=$t python tools/bench_conv.py --only "$ONLY" --iters 5 | grep -v weighted; done; done
for cus in 256 128; do for t in 0 64; do echo "== duo, workgroups = 2 x $cus, TUNE=$t"; KDCC_CONV_DUO=2 KDCC_PERSIST_CUS=$cus KDCC_CONV_TUNE=$t python tools/bench_conv.py --only "mod2 3x3 128,mod4 3x3 512" --iters 5 | grep -v weighted; done; done
