echo "=== C: the edit tests of test_gpu_parity in file order (3 x)"
for k in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edit_distance" 2>&1 | grep -v "^$" | tail -2; done
bash tools/gpu_r06.sh suite r06P
grep -n "FAILED\|ERROR" gpurun_out/r06P_pytest.txt | head
