import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d["recall_at_k"])
for k,v in d["legs"]["c4"].items():
    if isinstance(v,dict): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_forward","ms_per_forward_mean","TFLOPs","ms_per_256_queries")})
print({k:(round(v["ms_per_batch"],4)) for k,v in d["legs"].items() if "ms_per_batch" in v})
print({k:round(v.get("streaming_kernels_ms",v.get("staging_kernels_ms")),3) for k,v in d["legs"]["c5"].items() if isinstance(v,dict)})
for k, v in d["legs"].items():
    if "ms_per_batch_segments" in v:
        print(k, "segments", [round(x, 4) for x in v["ms_per_batch_segments"]], "fallback", v["exact_fallback_queries"], "lists", v.get("n_from_lists"),
              "dense", v.get("n_dense_exact"), "overfetch", v.get("overfetch"), "scan", round(v["scan_ms_per_launch"], 4), v.get("segment_diagnostics"))
print("cpu", d.get("cpu_baseline"))
print("facade", d.get("facade"))
