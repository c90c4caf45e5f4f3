#!/bin/bash
# threaded hyper-parameter fits with several factories alive in one process, default hardware queues and 8
for q in "" 8; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  export GPU_MAX_HW_QUEUES=$q; [ -z "$q" ] && unset GPU_MAX_HW_QUEUES
  python tools/bench_hyper_fit.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['N'], 'scipy %.2f ms' % d['fmin_l_bfgs_b_ms'], 'device %.2f ms' % d['device_ms'])"
  python tools/bench_latency.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'gpu_fit_optimised_ms' in d: print('latency tool N', d['N'], 'scipy %.2f' % d['gpu_fit_optimised_ms'], 'device %.2f' % d['gpu_fit_optimised_device_ms'])"
done
