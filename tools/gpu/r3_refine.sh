set -u
O=gpurun_out/r3r
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "on_device or gradient_stage or one_launch_optimiser or refine" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
