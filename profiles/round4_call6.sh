set +e
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_loop_gpu.py tests/test_demo_share_gpu.py tests/test_plan_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/r4_pytest_new.txt 2>&1; tail -6 gpurun_out/r4_pytest_new.txt | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash profiles/runtime_knobs_r4.sh
