for shape in "256 56 56 64 256 1 1 0 --residual --relu" "256 56 56 256 64 1 1 0 --relu" "256 28 28 128 512 1 1 0 --residual --relu" "256 28 28 512 128 1 1 0 --relu" "256 28 28 128 128 3 1 1 --relu" "256 14 14 256 1024 1 1 0 --residual --relu" "256 14 14 1024 256 1 1 0 --relu" "256 14 14 256 256 3 1 1 --relu" "256 7 7 512 512 3 1 1 --relu" "256 7 7 512 2048 1 1 0 --residual --relu"; do
  echo "== $shape"; echo "-- head"; RN_HIP_LIB=$PWD/.ab/librn_hip_head.so python tools/conv_bench.py $shape --reps 20 2>&1 | grep -v amdgpu.ids | grep -E "128x64|64x128"
  echo "-- new"; python tools/conv_bench.py $shape --reps 20 2>&1 | grep -v amdgpu.ids | grep -E "128x64|64x128|auto"
done
