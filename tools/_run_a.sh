timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -20
