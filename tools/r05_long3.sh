#!/bin/bash
python3 -m pytest tests -m gpu -x -q -k "long or attention or chat or 4b or config3 or prefill" 2>&1 | tail -3
for rep in 1 2; do echo "new: $(python3 tools/longctx_prof.py qwen3-4b 2300 32 2>/dev/null | head -3 | tr '\n' ' ')"; done
