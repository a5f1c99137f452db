#!/bin/bash
# usage (on the GPU box): tools/calib_write_size.sh <round tag>  ->  gpurun_out/<tag>_write_size_calibration.json
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/calib_$tag
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $out/calib $R/tools/calib_write_size.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -o p -- $out/calib > $out/w.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- $out/calib > $out/f.log 2>&1
cd $R
python - $out $tag <<'PY'
import csv, glob, json, sys
d, tag = sys.argv[1], sys.argv[2]
N = 8 << 20
rows = {}
for which in ("w", "f"):
    for f in glob.glob(d + "/" + which + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.setdefault((r["Kernel_Name"].split("(")[0], int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = {"_comment": "WRITE_SIZE / FETCH_SIZE (KiB) of kernels with known traffic, tools/calib_write_size.hip, 8 Mi threads each",
       "kernels": []}
expect = {"k_store16": ("16 B per thread, coalesced", 16 * N), "k_store32": ("32 B per thread, coalesced", 32 * N),
          "k_store4_strided": ("4 B per thread at a 32 B stride", 4 * N)}
atom = 0
for (k, disp), c in sorted(rows.items(), key=lambda kv: kv[0][1]):
    e = {"kernel": k, "dispatch": disp, "WRITE_SIZE_KiB": c.get("WRITE_SIZE"), "FETCH_SIZE_KiB": c.get("FETCH_SIZE")}
    if k in expect:
        e["what"], e["payload_bytes"] = expect[k]
        e["write_bytes_counted_per_payload_byte"] = round(1024 * c.get("WRITE_SIZE", 0) / expect[k][1], 4)
    elif k == "k_atomic64":
        e["what"] = "8 Mi device-scope 64-bit atomicAdd over %s addresses" % ("128000 (1 MB)" if atom == 0 else "8 Mi (64 MiB)")
        e["write_bytes_counted_per_atomic"] = round(1024 * c.get("WRITE_SIZE", 0) / N, 3)
        e["read_bytes_counted_per_atomic"] = round(2 * 1024 * c.get("FETCH_SIZE", 0) / N, 3)
        atom += 1
    out["kernels"].append(e)
path = d + "/../%s_write_size_calibration.json" % tag
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
