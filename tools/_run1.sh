python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/ab.sh head 3
