O=gpurun_out/occ; mkdir -p $O
for hi in 0 1; do
  VR_X3H_HI=$hi timeout 80 python bench.py --mode infer --steps 10 --warmup 2 --no-cpu-baseline > $O/i$hi.json 2>$O/i$hi.err
  VR_X3H_HI=$hi timeout 80 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/t$hi.json 2>$O/t$hi.err
done
VR_WGRAD_X3H=1 timeout 80 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/tw.json 2>$O/tw.err
timeout 200 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -q -s -k "three_fp16 or split_bf16_mode or fused_upsample" > $O/pytest.log 2>&1; echo rc=$? >> $O/pytest.log
