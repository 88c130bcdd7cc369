# tuning-build ablations of conv_row_tall_kernel (GPU box, from the repo root): KDCC_CONV_TUNE 64 = no epilogue, 128 = no output store,
# 2048 = every tile reads the same few input rows (L2-resident A operand)
export KDCC_LIB=tuning KDCC_BENCH_BATCH=8
for t in 0 64 2048 2112; do echo "== tall TUNE=$t"; KDCC_CONV_TUNE=$t python tools/bench_conv.py --only "mod2 3x3 128" --iters 10 | grep -v weighted; done
