#!/bin/bash
# full GPU suite + a bench line (end-of-change check)
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
