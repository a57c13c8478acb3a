#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc.sh for one kernel into a JSON file for profiles/.

    python tools/pmc_summary.py gpurun_out/<name> <kernel-substring> <workload> <frames> > profiles/rNN_pmc_<...>.json

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE come in KiB from separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced streaming read, so it is doubled."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, kernel, workload, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    agg = collections.defaultdict(lambda: [0, 0.0])
    names = set()
    for path in sorted(glob.glob(os.path.join(out_dir, "p*/*/*counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                name = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "")
                names.add(name.split("(")[0])          # template arguments kept, the parameter list dropped
                a = agg[r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    counters = {k: v[1] / v[0] for k, v in agg.items()}
    import bench
    system, desc = bench.make_system(workload)
    res = {
        "kernel": sorted(names), "workload": f"{workload}: {desc}, {frames} frames per launch",
        "collected_with": "tools/pmc.sh (separate rocprofv3 --kernel-trace --pmc passes; mean per dispatch)",
        "dispatches": {k: v[0] for k, v in agg.items()}, "counters": counters,
        "algorithmic_bytes_per_launch": system.bytes_per_frame * frames,
    }
    if "FETCH_SIZE" in counters:
        res["hbm_traffic_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters.get("WRITE_SIZE", 0.0)) * 1024.0
        res["traffic_over_algorithmic"] = res["hbm_traffic_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
        res["note"] = ("FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reports half of the bytes of a wide "
                       "coalesced streaming read (MI355X_MICROARCH.md, HBM) -> doubled")
    if "SQ_INSTS_VALU" in counters and "GRBM_GUI_ACTIVE" in counters:
        # 1024 SIMD-32 units; a wave64 VALU instruction occupies its SIMD for 2 cycles when other waves fill the gaps
        # (MI355X_MICROARCH.md: 'issues each VALU instruction over 2 cycles'; 4 is what ONE wave alone sustains);
        # GRBM_GUI_ACTIVE sums the 8 XCDs.  (Round 2 used 4 cycles here and reported fractions above 1.)
        res["valu_busy_fraction"] = counters["SQ_INSTS_VALU"] * 2.0 / 1024.0 / (counters["GRBM_GUI_ACTIVE"] / 8.0)
        res["valu_instructions_per_wave"] = counters["SQ_INSTS_VALU"] / counters["SQ_WAVES"] if counters.get("SQ_WAVES") else None
        # what the busy fraction above cannot say: how many SIMD cycles the kernel spends per VALU instruction it issues.
        # tools/microbench/valu_rate.hip on this chip, 8 waves per SIMD: 2.8 cycles for independent v_fma_f32, 4.3 for a
        # dependent chain and for v_pk_fma_f32 (two fmas each) — a kernel between those figures is bound by VALU issue.
        res["simd_cycles_per_valu_instruction"] = 1024.0 * (counters["GRBM_GUI_ACTIVE"] / 8.0) / counters["SQ_INSTS_VALU"]
        res["valu_rate_reference"] = {"independent_v_fma_f32": 2.8, "dependent_v_fma_f32": 4.3, "v_pk_fma_f32": 4.3,
                                      "source": "tools/microbench/valu_rate.hip, 8 waves per SIMD"}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
