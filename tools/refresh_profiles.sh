#!/bin/bash
# Run on the GPU box (gpurun): regenerates every measured artefact of profiles/ for this round into gpurun_out/prof/.
# Copy the files from gpurun_out/prof/ into profiles/ afterwards (rN_ prefix is added here).
#   1 headline bench + rocprofv3 --kernel-trace --stats of the same command
#   2 HBM traffic of k_scan from the PMC counters (separate --pmc passes, MI355X_MICROARCH.md §HBM) + the sha256 of the
#     kernel source it was measured on (bench.py refuses a stale file)
#   3 shard-size lines (1 M and 1.25 M rows), the other BASELINE.json configs (tools/bench_configs.py), C1 timing
#   4 encoder: kernel stats of one bge-base 256 x 64 forward loop + MFMA-busy PMC of the FFN-up GEMM
#   5 N-array fusion (C5) kernel stats
set -e
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
O=$R/gpurun_out/prof
STEPS=${STEPS:-12345}   # which parts to run (a gpurun call is limited to 20 minutes: e.g. STEPS=12, then STEPS=345)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [[ $STEPS == *1* ]]; then
echo "[1] bench" >&2
python3 $R/bench.py > $O/${ROUND}_bench_1gpu.json 2> $O/bench.err
# (the profiled run leaves out the extra legs, the CPU legs and the facade leg: every k_scan<false,...> launch in the stats is a
# 10 M-row launch of the headline workload, so the kernel's average there is comparable with roofline.ms_per_launch)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -- python3 $R/bench.py --no-cpu --recall-queries 0 --no-facade --no-legs > $O/bench_prof.json 2> $O/bench_prof.err
cp $(ls $O/kstats/*/*kernel_stats.csv | head -1) $O/${ROUND}_bench_10m_1gpu_kernel_stats.csv
rm -rf $O/kstats
# the same command with ONE batch in flight: scans never overlap, so rocprofv3's average k_scan<false,...> duration IS the kernel's
# time (with three in flight the next scan's workgroups take over CU by CU and the profiler's span of a scan includes that wait)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats1 -- python3 $R/bench.py --in-flight 1 --no-cpu --recall-queries 0 --no-facade --no-legs > $O/bench_prof_serial.json 2> $O/bench_prof_serial.err
cp $(ls $O/kstats1/*/*kernel_stats.csv | head -1) $O/${ROUND}_bench_10m_1gpu_in_flight_1_kernel_stats.csv
rm -rf $O/kstats1
fi
if [[ $STEPS == *2* ]]; then
echo "[2] pmc traffic" >&2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/scan_perf.py --rows 10000000 --steps 4 --mode sync > $O/pmc_$c.log 2>&1
done
python3 - "$R" "$O" "$ROUND" <<'PY'
import csv, glob, hashlib, json, sys
R, O, ROUND = sys.argv[1:4]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{O}/pmc_{c}/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_scan<false" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = [float(r["Counter_Value"]) for r in rows]
    kernel = rows[0]["Kernel_Name"]
    with open(f"{O}/{ROUND}_pmc_{c}_k_scan.csv", "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
fetch = sum(out["FETCH_SIZE"]) / len(out["FETCH_SIZE"]); write = sum(out["WRITE_SIZE"]) / len(out["WRITE_SIZE"])
rows_n, dim = 10_000_000, 768
bpv = 1.5 if ", true, true>" in kernel else 2.0   # k_scan<DENSE, CH, NT, STREAM, F12>: the 12-bit image
traffic = (2 * fetch + write) * 1024
sys.path.insert(0, f"{R}/tools")
from scan_source_hash import scan_source_sha256
json.dump({"kernel": kernel.replace("void anr::", "").split("(")[0], "rows": rows_n, "dim": dim,
           "bytes_per_stored_value": bpv,
           "algorithmic_bytes_per_launch": rows_n * dim * bpv, "FETCH_SIZE_KiB_per_launch": fetch,
           "WRITE_SIZE_KiB_per_launch": write,
           "correction": "gfx950: FETCH_SIZE counts half of a wide coalesced stream -> doubled (MI355X_MICROARCH.md HBM section); WRITE_SIZE exact; unit KiB",
           "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": traffic / (rows_n * dim * bpv),
           "kernel_source_sha256": scan_source_sha256(), "kernel_source_region": "k_scan",
           "command": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python3 tools/scan_perf.py --rows 10000000 --steps 4 --mode sync (one pass per counter; tools/refresh_profiles.sh)"},
          open(f"{O}/{ROUND}_pmc_traffic_k_scan.json", "w"), indent=1)
print("traffic/algorithmic", traffic / (rows_n * dim * bpv), "bytes per value", bpv)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
fi
if [[ $STEPS == *3* ]]; then
echo "[3] shard sizes + configs" >&2
{ python3 $R/tools/scan_perf.py --rows 1000000 --steps 60; python3 $R/tools/scan_perf.py --rows 1250000 --steps 60; python3 $R/tools/scan_perf.py --rows 1250000 --steps 60 --clustered; } 2>&1 | grep -v amdgpu > $O/${ROUND}_shard_size_lines.txt
# the pipeline's own time line (batch log) at the shard size: default, with the round-3 completion marker, shadow kernels, role streams
{ for cfg in "0 0 0" "0 1 0" "0 0 2" "1 0 0" "1 0 2"; do set -- $cfg; python3 $R/tools/pipeline_log.py --rows 1250000 --steps 200 --schedule $1 --stream-wait $2 --shadow $3; done; python3 $R/tools/pipeline_log.py --rows 1250000 --steps 60 --clustered --sigma 0.02; } 2>&1 | grep -v amdgpu > $O/${ROUND}_pipeline_log_shard_1250k.txt
python3 $R/tools/bench_configs.py > $O/configs.log 2>&1 && cp $R/gpurun_out/configs.json $O/${ROUND}_configs_c1_c2_c4_c5.json
{ python3 $R/tools/tiny_perf.py; echo "--- K=32"; K=32 python3 $R/tools/tiny_perf.py; echo "--- 20k x 768"; ROWS=20000 DIM=768 python3 $R/tools/tiny_perf.py; echo "--- 100k x 768"; ROWS=100000 DIM=768 python3 $R/tools/tiny_perf.py; } 2>&1 | grep -v amdgpu > $O/${ROUND}_c1_tiny_path.txt
fi
if [[ $STEPS == *4* ]]; then
echo "[4] encoder" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc -- python3 $R/tools/enc_perf.py 256 64 > $O/enc_perf.log 2>&1
cp $(ls $O/enc/*/*kernel_stats.csv | head -1) $O/${ROUND}_encoder_bge_base_256x64_kernel_stats.csv
python3 $R/tools/role_times.py $O/enc > $O/${ROUND}_encoder_role_times_fused.txt 2>&1 || true
rm -rf $O/enc
{ python3 $R/tools/enc_perf.py 256 64; python3 $R/tools/enc_perf.py 64 512; SHAPE=bge-m3 python3 $R/tools/enc_perf.py 256 64; } 2>&1 | grep TFLOP > $O/${ROUND}_encoder_forward_lines.txt
{ for lanes in 2 1; do ANORAG_ENC_LANES=$lanes python3 $R/tools/shared_forward_perf.py 2>&1 | grep -E "lanes|threads"; done; } > $O/${ROUND}_shared_forward_lines.txt
python3 $R/tools/query_latency.py 2>&1 | grep -v "amdgpu\|Warning\|it/s" > $O/${ROUND}_query_latency.txt
bash $R/tools/pmc_kernel.sh "k_gemm_pp<1" SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY TCC_HIT_sum TCC_MISS_sum -- $R/tools/enc_perf.py 256 64 > $O/${ROUND}_pmc_k_gemm_pp_ffn_up.txt 2>&1
fi
if [[ $STEPS == *5* ]]; then
echo "[5] fusion" >&2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fd -- python3 $R/tools/fuse_dense_perf.py 2>&1 | grep "queries x" > $O/${ROUND}_fuse_dense_c5.txt
cp $(ls $O/fd/*/*kernel_stats.csv | head -1) $O/${ROUND}_fuse_dense_c5_kernel_stats.csv
rm -rf $O/fd
{ python3 $R/tools/bm25_fuse_perf.py; echo "--- rare-term queries (vocabulary ranks 1000..30000)"; QLO=1000 QHI=30000 python3 $R/tools/bm25_fuse_perf.py; } 2>&1 | grep -v amdgpu > $O/${ROUND}_bm25_fuse_pipeline_lines.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bm -- python3 $R/tools/bm25_fuse_perf.py > /dev/null 2>&1
cp $(ls $O/bm/*/*kernel_stats.csv | head -1) $O/${ROUND}_bm25_fuse_common_terms_kernel_stats.csv
rm -rf $O/bm
python3 $R/tools/gemm_yardstick.py 2>&1 | grep -v amdgpu > $O/${ROUND}_gemm_yardstick.txt
fi
ls -la $O >&2
