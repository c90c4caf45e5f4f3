# every padded size class Np = 256 ... 12288 (N = Np - 37), four fits each: one value per size, and the likelihood under the
# default outer block against TGP_OB=512 (found: blocks of 1024 with a last block of 768).  gpurun -- bash tools/gpu/r4_ladder.sh
OFF=${OFF:-37}
SZ=$(python3 -c "print(' '.join(str(n-$OFF) for n in range(256, 12289, 256)))")
timeout -k 10 500 python tools/repeat_fit.py $SZ --trials 4 > gpurun_out/ladder_default.txt 2> gpurun_out/ladder_default.err; echo "default rc=$?"
TGP_OB=512 timeout -k 10 500 python tools/repeat_fit.py $SZ --trials 4 > gpurun_out/ladder_ob512.txt 2> gpurun_out/ladder_ob512.err; echo "ob512 rc=$?"
python3 - <<'PY'
import json
a=[json.loads(l) for l in open('gpurun_out/ladder_default.txt') if l.startswith('{')]
b=[json.loads(l) for l in open('gpurun_out/ladder_ob512.txt') if l.startswith('{')]
print(len(a), len(b))
worst=0
for x,y in zip(a,b):
    assert x['N']==y['N']
    r=abs(float(x['lml'])-float(y['lml']))/abs(float(x['lml']))
    worst=max(worst,r)
    if r>1e-10 or x['disagreeing'] or y['disagreeing']: print('DIFF', x['N'], x['lml'], y['lml'], r, x['disagreeing'], y['disagreeing'])
print('worst rel diff of LML between OB choices', worst)
PY
tail -3 gpurun_out/ladder_default.err
