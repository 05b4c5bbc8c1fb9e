# parity tests of the raster / shading path, then the bench line of every librdr_hip*.so in renderer_amd/lib
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_skinning.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/ab_tests.log 2>&1
for path in renderer_amd/lib/librdr_hip*.so; do
  lib=$(basename $path)
  RDR_HIP_LIB=$PWD/$path timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$lib.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$lib.json"))
print("$lib", round(d["value"],1), {k: round(v, 3) for k, v in d["device_ms"].items()})
PY
done
