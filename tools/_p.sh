timeout 900 python -m pytest tests/test_gpu_train_mlp.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/train_op_profile.py > gpurun_out/train_op_profile.txt 2>&1
grep -E "10 steps|aten::copy_|hipMemcpy|aten::nonzero|aten::mm|aten::bmm|aten::addmm|hipLaunchKernel|hipDeviceSync" gpurun_out/train_op_profile.txt | cut -c1-250 | head -30
python bench.py --workload train --steps 20 2>/dev/null | tail -1 | cut -c1-300
