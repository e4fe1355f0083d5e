#!/usr/bin/env python3
"""gpurun_out/prof/<workload>/ (tools/profile_round.sh) -> profiles/rNN/ + profiles/pmc_per_launch.json.

Per workload: kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command), the bench
lines (plain, under rocprof), pmc_summary.json = mean per launch of every counter per kernel, and
the figures bench.py quotes, stamped with the library's build id and the commit:
  hbm_bytes_per_launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024   (KB units; FETCH doubled per the
      gfx950 correction of MI355X_MICROARCH.md; separate --pmc passes) of the dominant kernel
  valu_busy_frac       = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
  valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
A bench line whose config.build_id is not the build id of the sources in this tree is REFUSED (exit status 2): figures are
quoted for the final build only.
usage: python tools/stamp_profiles.py r02 [gpurun_out/prof]
       python tools/stamp_profiles.py --check profiles/r05     every log / json of the directory that names a library build
                                                               ("library build <id>", "build <id>", "build_id": "<id>") must name THIS one;
                                                               files listed in <dir>/OTHER_BUILDS (one name per line, with the reason) are
                                                               A/B records of other builds and are skipped"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (template arguments: MODE, FILT, ANYSRC, T, PROG, SCAT, RNG (1 strict libm emitters, 2 53-bit draws), SCHED, NOBIN)
DOMINANT = {"point1e7": "trace_queue_kernel<0, true, false, double, 1, false, 0, 0, false>",
            "ring1e8": "trace_queue_kernel<0, true, false, double, 2, false, 0, 0, false>",
            "full1e9": "trace_queue_kernel<0, true, false, double, 1, false, 0, 0, false>"}
# the informational legs of a workload: their program kernels (bench.py `fp32` / `fast_fp64`)
LEGS = {"point1e7": {"fp32": "trace_queue_kernel<0, true, false, float, 1, false, 0, 0, false>",
                     "fast_fp64": "trace_queue_kernel<0, true, false, ort::fastd, 1, false, 0, 0, false>",
                     "strict": "trace_queue_kernel<0, true, false, double, 1, false, 1, 0, false>",
                     "wide": "trace_queue_kernel<0, true, false, double, 1, false, 2, 0, false>",
                     "strict_wide": "trace_queue_kernel<0, true, false, double, 1, false, 3, 0, false>"}}


def short(name):
    m = re.search(r"(trace_queue_kernel|trace_batch_kernel|trace_batch_rerun_kernel|trace_kernel|scatter_front_kernel|bin_log_kernel|fold_slabs_kernel|fold_kernel|emit_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def check(directory):
    from opticalraytrace_amd import capi
    build = capi.source_build_id()
    skip = set()
    other = os.path.join(directory, "OTHER_BUILDS")
    if os.path.exists(other):
        skip = {ln.split()[0] for ln in open(other).read().splitlines() if ln.strip() and not ln.startswith("#")}
    bad, seen = [], 0
    for root, _, files in os.walk(directory):
        for f in sorted(files):
            rel = os.path.relpath(os.path.join(root, f), directory)
            if rel in skip or f == "OTHER_BUILDS" or not f.endswith((".log", ".json", ".csv", ".txt")):
                continue
            text = open(os.path.join(root, f), errors="replace").read()
            ids = set(re.findall(r'(?:library build|^build|"build_id":)\s*"?([0-9a-f]{16})', text, re.M))
            seen += bool(ids)
            if ids - {build}:
                bad.append((rel, sorted(ids - {build})))
    for rel, ids in bad:
        print(f"REFUSED {rel}: taken on build {', '.join(ids)}, the sources here are {build}")
    print(f"{directory}: {seen} files name a build, {len(bad)} of them another one than {build}")
    return 2 if bad else 0


def main():
    if sys.argv[1] == "--check":
        sys.exit(check(sys.argv[2]))
    rnd = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof")
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    from opticalraytrace_amd import capi
    build = capi.source_build_id()
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    stamped = {"build_id": build, "commit_at_capture": commit, "round": rnd, "workloads": {},
               "source": "rocprofv3 --pmc passes of tools/profile_round.sh (separate runs per counter set), means per launch; "
                         "FETCH_SIZE / WRITE_SIZE in KB, FETCH doubled (gfx950 correction, MI355X_MICROARCH.md)"}
    for w in sorted(os.listdir(src)):
        d = os.path.join(src, w)
        if not os.path.isdir(d) or w not in DOMINANT:
            continue
        for f, name in (("bench.json", f"{w}_bench.json"), ("bench_under_rocprof.json", f"{w}_bench_under_rocprof.json")):
            if os.path.exists(os.path.join(d, f)):
                lines = [ln for ln in open(os.path.join(d, f)).read().splitlines() if ln.startswith("{")]
                if lines:
                    line = json.loads(lines[-1])
                    if line.get("config", {}).get("build_id") != build:
                        print(f"REFUSED {os.path.join(d, f)}: bench line of build {line.get('config', {}).get('build_id')}, the sources here are {build}")
                        sys.exit(2)
                    json.dump(line, open(os.path.join(dst, name), "w"), indent=1)
        ks = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
        if ks:
            shutil.copy(ks[0], os.path.join(dst, f"{w}_kernel_stats.csv"))
        # launches of full size only: bench.py's set-up also runs each kernel once on 64 rays
        rows = []
        for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
            rows += [r for r in csv.DictReader(open(f)) if short(r["Kernel_Name"])]
        biggest = collections.defaultdict(int)
        for r in rows:
            biggest[short(r["Kernel_Name"])] = max(biggest[short(r["Kernel_Name"])], int(r["Grid_Size"]))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = short(r["Kernel_Name"])
            if int(r["Grid_Size"]) == biggest[k]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summ = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
                for k, cs in sorted(agg.items())}
        json.dump(summ, open(os.path.join(dst, f"{w}_pmc_summary.json"), "w"), indent=1)
        dom = summ.get(DOMINANT[w])
        if not dom:
            print(f"{w}: dominant kernel {DOMINANT[w]} not in the PMC output: {list(summ)}")
            continue
        g = lambda c: dom.get(c, {}).get("mean_per_launch")          # noqa: E731
        bench = json.load(open(os.path.join(dst, f"{w}_bench.json"))) if os.path.exists(os.path.join(dst, f"{w}_bench.json")) else {}
        cfg = bench.get("config", {})
        entry = {"kernel": DOMINANT[w], "rays_per_launch_nominal": cfg.get("rays_per_gpu_per_launch")}
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            entry["fetch_bytes_per_launch"] = 2.0 * g("FETCH_SIZE") * 1024.0
            entry["write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0
            entry["hbm_bytes_per_launch"] = entry["fetch_bytes_per_launch"] + entry["write_bytes_per_launch"]
        if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
            entry["valu_busy_frac"] = g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / (g("GRBM_GUI_ACTIVE") / 8.0)
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            entry["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
        for c, key in (("SQ_INSTS_VALU", "valu_instructions_per_launch"), ("SQ_INSTS_SALU", "salu_instructions_per_launch"),
                       ("SQ_WAVES", "waves_per_launch"), ("TCC_EA0_ATOMIC_sum", "memory_side_atomics_per_launch"),
                       ("SQ_INSTS_VALU_FMA_F64", "fma_f64"), ("SQ_INSTS_VALU_MUL_F64", "mul_f64"), ("SQ_INSTS_VALU_ADD_F64", "add_f64"),
                       ("SQ_INSTS_VALU_TRANS_F64", "trans_f64"), ("SQ_INSTS_VALU_INT32", "int32"), ("SQ_INSTS_VALU_INT64", "int64"),
                       ("SQ_INSTS_VALU_CVT", "cvt")):
            if g(c) is not None:
                entry[key] = g(c)
        if all(k in entry for k in ("fma_f64", "mul_f64", "add_f64")):
            # COUNTED fp64 flop of one launch: wave-instructions x 64 lanes, an FMA = 2 (the roofline's own convention)
            entry["counted_fp64_flop_per_launch"] = 64.0 * (entry["add_f64"] + entry["mul_f64"] + 2.0 * entry["fma_f64"])
            if entry.get("valu_instructions_per_launch"):
                entry["non_arithmetic_share_of_valu"] = 1.0 - (entry["add_f64"] + entry["mul_f64"] + entry["fma_f64"]) / entry["valu_instructions_per_launch"]
        # intersections per launch of the dominant kernel, from the bench line of the PMC-sized run
        p1 = os.path.join(d, "pmc1.json")
        if os.path.exists(p1):
            lines = [ln for ln in open(p1).read().splitlines() if ln.startswith("{")]
            if lines:
                b = json.loads(lines[-1])
                c2 = b["config"]
                per_step = c2["intersections_per_step"]
                if w == "full1e9":       # two phases per step: the point layer holds 6.31 of the 7.87 intersections per ray pair
                    per_step *= 6.3136 / (6.3136 + 1.5524)
                launches = max(1, -(-c2["rays_per_gpu_per_launch"] // (1 << 27)))     # ort_trace cuts at 2^27 rays
                entry["intersections_per_launch"] = per_step / launches
        stamped["workloads"][w] = entry
        print(w, json.dumps(entry, indent=1))
        for leg, kname in LEGS.get(w, {}).items():
            lk = summ.get(kname)
            if not lk:
                print(f"{w}: leg {leg}: kernel {kname} not in the PMC output")
                continue
            gl = lambda c: lk.get(c, {}).get("mean_per_launch")      # noqa: E731
            e2 = {"kernel": kname, "intersections_per_launch": entry.get("intersections_per_launch")}
            if gl("SQ_ACTIVE_INST_VALU") and gl("GRBM_GUI_ACTIVE"):
                e2["valu_busy_frac"] = gl("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / (gl("GRBM_GUI_ACTIVE") / 8.0)
            if gl("SQ_THREAD_CYCLES_VALU") and gl("SQ_ACTIVE_INST_VALU"):
                e2["valu_lane_utilisation"] = gl("SQ_THREAD_CYCLES_VALU") / (64.0 * gl("SQ_ACTIVE_INST_VALU"))
            for c, key in (("SQ_INSTS_VALU", "valu_instructions_per_launch"), ("SQ_INSTS_SALU", "salu_instructions_per_launch"),
                           ("SQ_WAVES", "waves_per_launch"), ("SQ_WAIT_INST_ANY", "wave_cycles_waiting_for_issue"),
                           ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_INSTS_VALU_FMA_F32", "fma_f32"), ("SQ_INSTS_VALU_MUL_F32", "mul_f32"),
                           ("SQ_INSTS_VALU_ADD_F32", "add_f32"), ("SQ_INSTS_VALU_TRANS_F32", "trans_f32"),
                           ("SQ_INSTS_VALU_INT32", "int32"), ("SQ_INSTS_VALU_FMA_F64", "fma_f64"), ("SQ_INSTS_VALU_MUL_F64", "mul_f64"),
                           ("SQ_INSTS_VALU_ADD_F64", "add_f64"), ("SQ_INSTS_VALU_TRANS_F64", "trans_f64")):
                if gl(c) is not None:
                    e2[key] = gl(c)
            stamped["workloads"][f"{w}_{leg}"] = e2
            print(w, leg, json.dumps(e2, indent=1))
    json.dump(stamped, open(os.path.join(ROOT, "profiles", "pmc_per_launch.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
