#!/usr/bin/env python3
"""python tools/batch_time.py <shape> <B> [steps] [ctx] -- batched decode step timing (for rocprofv3 runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import batch_check as bc
bc.timing(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 64, int(sys.argv[4]) if len(sys.argv) > 4 else 2048)
