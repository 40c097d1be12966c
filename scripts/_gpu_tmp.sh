timeout 500 python -m pytest tests -m gpu -q -x -k "image_normalize or power_step" 2>&1 | tail -3
timeout 100 python scripts/imgnorm_time.py
