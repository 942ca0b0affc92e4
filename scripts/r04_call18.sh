#!/bin/bash
python -m pytest tests/test_gpu_shard.py -q -x 2>&1 | tail -15
