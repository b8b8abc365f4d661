"""Copy the summaries scratch/profile_round.sh left under gpurun_out/prof_<tag> into profiles/ and rebuild
profiles/traffic.json from their PMC passes.  Usage: python scratch/collect_profiles.py"""
import ast
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r03"
CASES = {  # tag -> (traffic.json key, bench steps)
    ROUND + "_cfg3": ("cfg3_65536_moving/65536/shallow-attached", 200),
    ROUND + "_cfg2": ("cfg2_4096_static/4096/shallow-attached", 200),
    ROUND + "_cfg5": ("cfg5_dense_720/4096/none", 100),
}
traffic = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB). "
           "gfx950 reports half the bytes of wide coalesced reads, so FETCH_SIZE is doubled (an upper bound here: the "
           "loads are 8 B per lane, for which the counter is uncalibrated). bench.py copies the entry that matches its "
           "workload into roofline.traffic. valu_insts = SQ_INSTS_VALU (wave instructions) per launch from the same summary; "
           "valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (SIMDs x the launch's cycles, GRBM_GUI_ACTIVE / 8 XCDs): the share of "
           "the device's VALU issue cycles the kernel used; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; lds_conflict_frac = "
           "SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS."}
for tag, (key, steps) in CASES.items():
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    short = tag.replace(ROUND + "_", "")
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{ROUND}_bench_{short}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "pmc.txt"), os.path.join(ROOT, "profiles", f"{ROUND}_bench_{short}_pmc.txt"))
    entry = {"source": f"profiles/{ROUND}_bench_{short}_pmc.txt (rocprofv3 --pmc, four separate passes over `bench.py "
             f"--no-cpu-baseline --spin-seconds 0 --steps {steps} --repeats 1`)", "steps": steps}
    raw = {}
    for l in open(os.path.join(src, "pmc.txt")):
        m = re.match(r"auv::(k_sense\w*|k_vessel)(<\d>)? (\{.*\}) n=", l)
        if not m:
            continue
        name = "k_sense" if m.group(1).startswith("k_sense") else "k_vessel"
        d = ast.literal_eval(m.group(3))
        e = entry.setdefault(name, {})
        raw.setdefault(name, {}).update(d)
        if "FETCH_SIZE" in d:
            e["fetch_kib"] = round(d["FETCH_SIZE"], 1)
        if "WRITE_SIZE" in d:
            e["write_kib"] = round(d["WRITE_SIZE"], 1)
        if "SQ_INSTS_VALU" in d and name == "k_sense":
            e["valu_insts"] = round(d["SQ_INSTS_VALU"], 1)
    for name, d in raw.items():
        e = entry[name]
        if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d:
            e["valu_busy"] = round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * d["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
            e["wait_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
        if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_ACTIVE_INST_LDS"):
            e["lds_conflict_frac"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_ACTIVE_INST_LDS"], 4)
    traffic[key] = entry
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=2)
print(json.dumps(traffic, indent=1)[:1500])
