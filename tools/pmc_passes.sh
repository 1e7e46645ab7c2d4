#!/bin/bash
# Collect PMC counters for the integrator kernel in separate rocprofv3 passes (<= 8 SQ counters each; FETCH_SIZE / WRITE_SIZE in
# passes of their own, as MI355X_MICROARCH.md prescribes) and write <outdir>/kernel_pmc.json — the per-ray hardware counts that
# bench.py's roofline block replays (copy it to profiles/roundN_kernel_pmc.json).
# Usage (on the GPU box, from the repo root): tools/pmc_passes.sh <outdir> [spp] [script + args instead of bench.py, e.g. "tools/bench_scenes.py --only mesh6 --spp 8"]
set -u
R=$PWD; OUT=$R/${1:-gpurun_out/pmc}; SPP=${2:-64}
CMD=${3:-"bench.py --steps 1 --warmup 0 --spp $SPP --no-cpu-baseline --no-psnr --no-configs"}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/$CMD > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
run sq2 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT
run mix1 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64
run mix2 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run tcc3 TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections, json, hashlib, os
out = "$OUT"
ctr = {}
dur = []
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if "k_render_pass" in r["Kernel_Name"]:
                agg[r["Counter_Name"]][0] += 1; agg[r["Counter_Name"]][1] += float(r["Counter_Value"])
        for k, v in agg.items():
            ctr[k] = v[1] / v[0]
            print(f"{k:28s} launches={v[0]} per_launch={v[1]/v[0]:.6g}")
    if d.rstrip("/").endswith("grbm"):
        for f in glob.glob(d + "*/*_kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "k_render_pass" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
line = None
for l in open(out + "/sq1.log"):
    if l.startswith("{") and "rays_per_launch" in l:
        line = json.loads(l)
if line and ctr.get("SQ_INSTS_VALU"):
    rays = line["roofline"]["rays_per_launch"]
    cyc = ctr["GRBM_GUI_ACTIVE"] / 8.0
    ms = sum(dur) / len(dur) if dur else None
    j = {"kernel": "k_render_pass<MODE_LDS,no-sun,no-alpha>", "workload": line["config"]["workload"] + f", {line['config']['spp_total']} spp in this launch",
         "rays_per_launch": rays, "counters_per_launch": {k: v for k, v in sorted(ctr.items())},
         "valu_insts_per_ray": ctr["SQ_INSTS_VALU"] / rays,
         "lane_utilisation": round(ctr["SQ_THREAD_CYCLES_VALU"] / (64.0 * ctr["SQ_INSTS_VALU"]), 4) if "SQ_THREAD_CYCLES_VALU" in ctr else None,
         "cycles_per_launch": cyc, "launch_ms_under_profiler": ms, "shader_clock_ghz": round(cyc / (ms * 1e6), 3) if ms else None,
         "valu_issue_frac_2cycle": round(ctr["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc), 4),
         "hbm_bytes_per_ray": (2.0 * ctr.get("FETCH_SIZE", 0) + ctr.get("WRITE_SIZE", 0)) * 1024.0 / rays,
         "hbm_note": "FETCH_SIZE x 2 (gfx950 counts wide coalesced reads at 1/2: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, KB -> bytes",
         "kernels_hip_sha256_16": hashlib.sha256(open("$R/distributed-path-tracer_amd/csrc/kernels.hip", "rb").read() + open("$R/distributed-path-tracer_amd/csrc/device_core.hpp", "rb").read()).hexdigest()[:16],
         "source": "tools/pmc_passes.sh (separate rocprofv3 --pmc passes of one launch)"}
    json.dump(j, open(out + "/kernel_pmc.json", "w"), indent=1)
    print("wrote", out + "/kernel_pmc.json")
PY
