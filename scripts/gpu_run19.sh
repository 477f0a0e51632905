timeout 1500 python -m pytest tests -m gpu -q --timeout 400 -s 2>&1 | grep -E "cn_|passed|failed|FAILED|Error" | tail -15
