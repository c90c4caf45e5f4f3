set -u
mkdir -p gpurun_out/la5
S="700 1024 2048 3000 4096 8192"
python3 tools/bench_fit.py $S --check > gpurun_out/la5/f64.jsonl 2>&1
python3 tools/bench_fit.py $S --check --dtype f32 > gpurun_out/la5/f32.jsonl 2>&1
TGP_PANEL=3 python3 tools/bench_fit.py $S --check > gpurun_out/la5/panel3.jsonl 2>&1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/la5/*.jsonl')):
    out=[]
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); out.append(f"{d['N']}:{d['fit_ms_device']:.3f}({d.get('lml_rel_err',0):.0e})")
    print(f.split('/')[-1], ' '.join(out))
PY
timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/la5/pytest.log 2>&1; tail -3 gpurun_out/la5/pytest.log
