. "$(dirname "$0")/common.sh"
set +e
bash tools/gpu_tests.sh r04m/tests "glass or pool or catcher or config4 or sky or one_shot or dispers or relaxed_arithmetic_stays or medium or thin or portal or image_light"
for scene in "--scene outdoor --width 3840 --height 2160 --steps 2" "--scene cornell_glass --steps 4"; do
  bash tools/ab_args.sh r04m 1 "$scene --split-nee 0" "$scene --split-nee 1"
  ZYG_AMD_LIB=$R/variants/lib_ew1.so bash tools/ab_args.sh r04m_ew1 1 "$scene --split-nee 1"
done
