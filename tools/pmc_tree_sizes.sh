OUT=gpurun_out/pmc_tree_sizes
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 16384 32768 65536 131072; do
  for pass in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${B}_$pass -- python3 tools/run_op.py atlas_like f32 rnea_grad $B > $OUT/${B}_$pass.log 2>&1 || echo "pass $B $pass failed"
  done
done
python3 - <<P
import csv, glob
for B in (16384, 32768, 65536, 131072):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob("$OUT/%d_%s/**/*counter_collection.csv" % (B, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "tree_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: v.append(float(r["Counter_Value"]))
        out[c] = sum(v) / max(1, len(v))
    alg = B * 7680
    print("B=%7d  alg %7.1f MB   FETCH_SIZE(raw, KB units x1024) %8.1f MB   WRITE_SIZE %8.1f MB   write/alg_out %.3f" % (B, alg / 1e6, out["FETCH_SIZE"] * 1024 / 1e6, out["WRITE_SIZE"] * 1024 / 1e6, out["WRITE_SIZE"] * 1024 / (B * 7320)))
P
