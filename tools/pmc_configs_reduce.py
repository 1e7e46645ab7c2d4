#!/usr/bin/env python3
"""Second half of tools/pmc_configs.sh: counters of the passes -> <outdir>/configs_pmc.json.  tools/pmc_configs_reduce.py <outdir> [renders per pass if the logs do not say]"""
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs
out = os.path.abspath(sys.argv[1])
res = {}
for cfg in ("config3_mesh82k_1080p_8b", "config4_atrium_1080p_8b", "config5_atrium_4k_16b", "jack_of_blades_1080p_8b"):
    line0 = None
    for l in open(f"{out}/{cfg}.sq1.log"):
        if l.startswith("{") and cfg in l: line0 = json.loads(l)[cfg]
    tag = "k_wf_traverse" if line0 and "queue" in line0.get("pipeline", "") else "k_render_pass"
    ctr = collections.defaultdict(float); n_launch = collections.defaultdict(int); dur = 0.0
    for f in glob.glob(f"{out}/{cfg}/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if tag in r["Kernel_Name"]:
                ctr[r["Counter_Name"]] += float(r["Counter_Value"]); n_launch[r["Counter_Name"]] += 1
    for f in glob.glob(f"{out}/{cfg}/g/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if tag in r["Kernel_Name"]: dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    line = None
    for l in open(f"{out}/{cfg}.sq1.log"):
        if l.startswith("{") and cfg in l: line = json.loads(l)[cfg]
    if not line or "SQ_INSTS_VALU" not in ctr: print("incomplete:", cfg); continue
    # every pass renders the scene several times (warm-up + timed frames; bench_configs says how many): the counters cover all, so do the rays
    renders = float(line.get("renders_in_process", sys.argv[2] if len(sys.argv) > 2 else 2))
    rays = renders * line["rays"]
    cyc = ctr["GRBM_GUI_ACTIVE"] / 8.0
    res[cfg] = {"kernel": tag, "workload": line["workload"], "rays_counted": rays, "launches": n_launch["SQ_INSTS_VALU"],
                "counters_total": dict(sorted(ctr.items())), "valu_insts_per_ray": ctr["SQ_INSTS_VALU"] / rays, "salu_insts_per_ray": ctr["SQ_INSTS_SALU"] / rays,
                "vmem_rd_insts_per_ray": ctr["SQ_INSTS_VMEM_RD"] / rays,
                "lanes_on": round(ctr["SQ_THREAD_CYCLES_VALU"] / (64.0 * ctr["SQ_INSTS_VALU"]), 4),
                "wait_frac": round(ctr["SQ_WAIT_ANY"] / ctr["SQ_WAVE_CYCLES"], 4), "issue_wait_frac": round(ctr["SQ_WAIT_INST_ANY"] / ctr["SQ_WAVE_CYCLES"], 4),
                "valu_issue_frac_2cycle_under_profiler": round(ctr["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc), 4),
                "kernel_ms_under_profiler": dur, "shader_clock_ghz": round(cyc / (dur * 1e6), 3) if dur else None,
                "l2_hit_rate": round(ctr["TCC_HIT_sum"] / (ctr["TCC_HIT_sum"] + ctr["TCC_MISS_sum"]), 4) if ctr.get("TCC_HIT_sum") else None,
                "hbm_bytes_per_ray": (2.0 * ctr.get("FETCH_SIZE", 0) + ctr.get("WRITE_SIZE", 0)) * 1024.0 / rays,
                "hbm_bytes_per_ray_fetch_uncorrected": (ctr.get("FETCH_SIZE", 0) + ctr.get("WRITE_SIZE", 0)) * 1024.0 / rays,
                "hbm_note": "FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 bytes; calibrated there for wide streaming reads — these kernels gather 16-byte pieces, so the uncorrected sum is given beside it as the lower bound)",
                "source_sha256_16": bench_configs.source_hash(), "renders_per_pass": renders, "source": "tools/pmc_configs.sh (separate rocprofv3 --pmc passes; each pass = warm-up + timed frames of the scene, all counted)"}
    print(cfg, json.dumps({k: v for k, v in res[cfg].items() if k != "counters_total"}))
json.dump({"configs": res}, open(out + "/configs_pmc.json", "w"), indent=1)
print("wrote", out + "/configs_pmc.json")
