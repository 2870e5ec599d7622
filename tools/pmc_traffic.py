"""HBM traffic per launch from the TCC counters (run ON the GPU box):  python tools/pmc_traffic.py
Two rocprofv3 passes per configuration (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace + --pmc only.
gfx950 correction (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE counts 128-B requests at 64 B -> doubled for our 16 B/lane
coalesced reads; WRITE_SIZE is taken as reported (uncalibrated).  Both counters are in KiB.
Writes gpurun_out/pmc_traffic.json: {"<kernel>:S<seq>:<mode>": {"fetch_bytes":…, "write_bytes":…, "bytes":…}}"""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"attn_fwd_kernel": "attn_fwd", "attn_fwd64_kernel": "attn_fwd", "attn_fwd_split_kernel": "attn_fwd", "attn_bwd_fused_kernel": "attn_bwd_fused", "attn_bwd_fused64_kernel": "attn_bwd_fused", "attn_bwd_kv64_mixed_kernel": "attn_bwd_dkdv", "attn_bwd_q_kernel": "attn_bwd_dq",
         "attn_bwd_kv_kernel": "attn_bwd_dkdv", "attn_bwd_kv64_kernel": "attn_bwd_dkdv", "attn_bwd_q64_kernel": "attn_bwd_dq", "drpe_reduce_kernel": "bias_grad_reduce", "attn_bwd_dbias_kernel": "attn_bwd_dbias", "dbias_reduce_kernel": "dbias_reduce",
         "attn_bwd_qdb64_kernel": "attn_bwd_dq", "attn_fwd64_dense_kernel": "attn_fwd", "attn_fwd64_mixed_kernel": "attn_fwd", "dbias_partial_reduce_kernel": "dbias_reduce",
         "attn_bwd_dfused64_kernel": "attn_bwd_fused", "bwd_stat2_kernel": "bwd_stat2", "attn_fwd64_w1_kernel": "attn_fwd"}  # (round 5: attn_bwd_dq of a dense problem = dQ + the batch-reduced dbias)
out = {}
for S, mode in ((512, "rpe"), (2048, "rpe"), (8192, "rpe"), (8192, "none"), (512, "dense"), (2048, "dense"), (8192, "dense")):
    vals = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = f"/tmp/pmc_{S}_{mode}_{ctr}"
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "tools", "run_one.py"), "--S", str(S), "--mode", mode, "--what", "both", "--iters", "3"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not files:
            print("no counter file", S, mode, ctr, r.stderr[-500:]); continue
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != ctr: continue
            for k, v in NAMES.items():
                if k in row["Kernel_Name"]:
                    acc[v].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            vals[k][ctr] = sum(v) / len(v)
    for k, v in vals.items():
        fb = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024.0
        wb = v.get("WRITE_SIZE", 0.0) * 1024.0
        out[f"{k}:S{S}:{mode}"] = {"fetch_bytes": fb, "write_bytes": wb, "bytes": fb + wb,
                                  "raw": {"FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "WRITE_SIZE_KiB": v.get("WRITE_SIZE")}}
        print(f"{k}:S{S}:{mode}: fetch {fb/1e6:.2f} MB (2x corrected)  write {wb/1e6:.2f} MB")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
