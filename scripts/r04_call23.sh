#!/bin/bash
python -m pytest tests/test_gpu_e2e.py -q 2>&1 | tail -2
DSAC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload config5 --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['train_step']; print('config5 2 ranks gloo:', j['n_gpus'], 'ranks', round(t['step_ms'],1), 'ms/step, collectives', t['collectives_per_step'], 'exposed', round(t['collective_exposed_ms'],1), 'loss', t['last_loss'])"
