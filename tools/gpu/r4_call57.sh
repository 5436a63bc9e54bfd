cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_index_gpu.py -x -q -m gpu -k "twelve_bit_scan_image" 2>&1 | grep -E "^E |passed|failed" | head -8 | cut -c1-900
