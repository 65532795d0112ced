#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_round.sh run from gpurun_out/<tag> into profiles/ (tracked) and refresh
profiles/pmc_traffic.json.  Usage: python tools/collect_round.py <tag> <profile-prefix, e.g. r01_j>"""
import collections, csv, glob, json, re, shutil, sys
tag, pre = sys.argv[1], sys.argv[2]
ks = glob.glob(f'gpurun_out/{tag}/prof/*/*kernel_stats.csv')[0]
shutil.copy(ks, f'profiles/{pre}_bench_kernel_stats.csv')
shutil.copy(f'gpurun_out/{tag}/bench.log', f'profiles/{pre}_bench.log')
rows = list(csv.DictReader(open(ks)))
steps = 12   # gpu_round.sh profiles bench.py --steps 5 --warmup 2 (+ 5 steps of the instrumented breakdown pass)
for r in rows[:30]:
    print(f"{r['Name'][:88]:88s} calls={r['Calls']:>5s} ms/step={float(r['TotalDurationNs'])/1e6/steps:8.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
print("total kernel ms/step", sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps, " kernels/step", sum(int(r['Calls']) for r in rows) / steps)
out = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'gpurun_out/{tag}/pmc_{C}/*/*counter_collection.csv')[0]
    shutil.copy(f, f'profiles/{pre}_pmc_{C}.csv')
    byd = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == C:
            byd[(r['Kernel_Name'], r['Dispatch_Id'])] += float(r['Counter_Value'])
    agg = collections.defaultdict(list)
    for (k, d), v in byd.items():
        name = re.split(r'[<(]', k.replace('void ', '').replace('(anonymous namespace)::', ''))[0]
        agg[name].append(v)
    for k, v in agg.items():
        out.setdefault(k, {})[C] = sum(v) / len(v)
for k, v in sorted(out.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', 0) + kv[1].get('WRITE_SIZE', 0)))[:6]:
    print(f"{k:32s} fetch_kib={v.get('FETCH_SIZE',0):12.0f} write_kib={v.get('WRITE_SIZE',0):12.0f} GB={(v.get('FETCH_SIZE',0)+v.get('WRITE_SIZE',0))*1024/1e9:.3f}")
try:      # memory-side atomic requests of the hash backward, mean per launch over the launches of the pass (= over the ray sets)
    f = glob.glob(f'gpurun_out/{tag}/pmc_TCC_EA0_ATOMIC_sum/*/*counter_collection.csv')[0]
    shutil.copy(f, f'profiles/{pre}_pmc_TCC_EA0_ATOMIC_sum.csv')
    byd = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'TCC_EA0_ATOMIC_sum' and 'hash_bwd_batched' in r['Kernel_Name']:
            byd[r['Dispatch_Id']] += float(r['Counter_Value'])
    atomic_requests = sum(byd.values()) / len(byd)
    print(f"hash backward: {atomic_requests:.0f} memory-side atomic requests per launch ({len(byd)} launches)")
except IndexError:
    atomic_requests = None
hb = out.get('hash_bwd_batched_kernel') or out.get('hash_bwd_cached_kernel') or out['hash_bwd_kernel']
hf = out['hash_fwd_kernel']
j = {"_comment": f"HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only), run {pre} (profiles/{pre}_pmc_*.csv). Counter values are KiB: bytes = (FETCH_SIZE + WRITE_SIZE) * 1024. gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 for wide (16 B/lane) streaming reads and is UNCALIBRATED for the 4-8 B gathers these kernels issue, so the fetch side is a lower bound; WRITE_SIZE is exact for float atomics and 16-B stores.",
     "lse_hash_fwd": {"fetch_kib": round(hf['FETCH_SIZE']), "write_kib": round(hf['WRITE_SIZE']), "bytes": round((hf['FETCH_SIZE'] + hf['WRITE_SIZE']) * 1024)},
     "lse_hash_bwd": {"kernel": "hash_bwd_batched_kernel<true,512,2>", "fetch_kib": round(hb['FETCH_SIZE']), "write_kib": round(hb['WRITE_SIZE']), "bytes": round((hb['FETCH_SIZE'] + hb['WRITE_SIZE']) * 1024)}}
if atomic_requests is not None:
    j["lse_hash_bwd"]["atomic_requests"] = round(atomic_requests)
try:      # keys other tools maintain in the same file (matrix_core_busy: tools/pmc_mlp.sh) survive a refresh of the traffic numbers
    old = json.load(open('profiles/pmc_traffic.json'))
    for k, v in old.items():
        if k not in j:
            j[k] = v
except OSError:
    pass
# stamp: the hash kernels' sources these traffic numbers were measured on (bench.py nulls them when the tree has moved on);
# the "mlp" stamp belongs to matrix_core_busy and is set by tools/collect_busy.py from the SQ-counter pass of the same tree
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location('prov', 'lsenerf_amd/provenance.py'); prov = importlib.util.module_from_spec(spec); spec.loader.exec_module(prov)
j.setdefault("source_digests", {})["hash"] = prov.source_digest("hash")
json.dump(j, open('profiles/pmc_traffic.json', 'w'), indent=1)
