cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_sharded_gpu.py -x -q -m gpu -k nccl 2>&1 | grep -E "Error|error|assert|^E |line" | head -30
