set -u
mkdir -p gpurun_out/la3
S="700 1024 1300 2048 3000 4096 8192"
TGP_BGINV=0 python3 tools/bench_fit.py $S --check > gpurun_out/la3/off.jsonl 2>&1
python3 tools/bench_fit.py $S --check > gpurun_out/la3/on_nomask.jsonl 2>&1
for c in 128 192 224; do
TGP_BG_CUS=$c python3 tools/bench_fit.py $S --check > gpurun_out/la3/on_$c.jsonl 2>&1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/la3/*.jsonl')):
    out=[]
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); out.append(f"{d['N']}:{d['fit_ms_device']:.3f}({d.get('lml_rel_err',0):.0e})")
        else: out.append(l.strip()[:80])
    print(f.split('/')[-1], ' '.join(out))
PY
