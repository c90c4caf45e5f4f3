set -u
mkdir -p gpurun_out/la4
S="512 700 1024 1300 1536 2048 2560"
for ob in 256 512; do
TGP_OB=$ob python3 tools/bench_fit.py $S --check > gpurun_out/la4/ob$ob.jsonl 2>&1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/la4/ob*.jsonl')):
    out=[]
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); out.append(f"{d['N']}:{d['fit_ms_device']:.3f}({d.get('lml_rel_err',0):.0e})")
    print(f.split('/')[-1], ' '.join(out))
PY
