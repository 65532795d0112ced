#!/usr/bin/env python3
"""matrix_core_busy of profiles/pmc_traffic.json from a tools/pmc_mlp.sh run of THIS tree: SQ_VALU_MFMA_BUSY_CYCLES summed over the
chip / (1024 SIMDs x kernel duration x clock), durations from the kernel trace of the same counter pass.
Usage: python tools/collect_busy.py <pmc out dir, e.g. gpurun_out/r4a/pmc_mlp> <run tag, e.g. r04_a> [clock GHz under profiling = 2.1]"""
import collections, csv, glob, importlib.util, json, sys
out, tag = sys.argv[1], sys.argv[2]
ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.1
busy, dur = collections.defaultdict(dict), collections.defaultdict(dict)
f = glob.glob(f'{out}/SQ_VALU_MFMA_BUSY_CYCLES/*/*counter_collection.csv')[0]
for r in csv.DictReader(open(f)):      # one row per (dispatch, counter [, dimension instance]): sum the rows of a dispatch
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES' and 'mlp_' in r['Kernel_Name']:
        k, d = r['Kernel_Name'], r['Dispatch_Id']
        busy[k][d] = busy[k].get(d, 0.0) + float(r['Counter_Value'])
        dur[k][d] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
busy = {k: list(v.values()) for k, v in busy.items()}
dur = {k: list(v.values()) for k, v in dur.items()}
res = {}
for k in busy:
    name = k.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
    b, t = sum(busy[k]) / len(busy[k]), sum(dur[k]) / len(dur[k])
    res[name] = round(b / (1024 * t * ghz), 3)
    print(f"{name:60s} busy cycles {b:14.0f}  duration {t / 1e3:8.1f} us  matrix-core busy {res[name]:.3f}")
spec = importlib.util.spec_from_file_location('prov', 'lsenerf_amd/provenance.py'); prov = importlib.util.module_from_spec(spec); spec.loader.exec_module(prov)
j = json.load(open('profiles/pmc_traffic.json'))
res["_comment"] = (f"SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x {ghz} GHz under profiling), durations from the kernel "
                   f"trace of the same counter pass; run {tag} (profiles/{tag}_pmc_sq_counters.txt)")
j["matrix_core_busy"] = res
j.setdefault("source_digests", {})["mlp"] = prov.source_digest("mlp")
json.dump(j, open('profiles/pmc_traffic.json', 'w'), indent=1)
