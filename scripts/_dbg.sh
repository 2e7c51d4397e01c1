mkdir -p gpurun_out/r04last
python scripts/debug_amp_identity.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r04last/amp_identity_default.txt
python scripts/debug_amp_identity.py 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r04last/amp_identity_default.txt
python -m pytest tests -m gpu -q > gpurun_out/r04last/pytest_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r04last/pytest_full.log | tee gpurun_out/r04last/pytest_gpu.txt
cat gpurun_out/r04last/amp_identity_default.txt
