# L2-miss traffic and time of the wide 1x1 / 3x3 layers against the N-tile group size of the persistent kernels (GPU box, repo root):
# one rocprofv3 --pmc FETCH_SIZE pass of tools/bench_conv.py per setting (FETCH_SIZE in KiB, x 2 on gfx950: MI355X_MICROARCH.md)
export KDCC_BENCH_BATCH=8 TMPDIR=/tmp
ONLY="1x1 2048->4096,1x1 4096->2048,1x1 1024->2048,mod7 3x3 d4 1024->2048"
for g in 4 2 8 16; do
  export KDCC_CONV_TNGROUP=$g
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/tng/g$g -o f -- python tools/bench_conv.py --only "$ONLY" --iters 3 > gpurun_out/tng/g$g.log 2>&1 || exit 1
  echo "== KDCC_CONV_TNGROUP=$g"; grep -v weighted gpurun_out/tng/g$g.log | grep TFLOP
  python - <<PY
import csv, collections
d = collections.defaultdict(lambda: [0.0, 0])
tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open("gpurun_out/tng/g$g/f_kernel_trace.csv"))} if False else {}
for r in csv.DictReader(open("gpurun_out/tng/g$g/f_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and ("persist" in r["Kernel_Name"] or "conv_row_lw" in r["Kernel_Name"]):
        k = (r["Kernel_Name"][28:70], r["Grid_Size"])
        d[k][0] += float(r["Counter_Value"]) * 2048; d[k][1] += 1
for k, (b, n) in d.items():
    print("   ", k, f"{b / n / 1e9:.3f} GB fetched per launch ({n} launches)")
PY
done
