# section profile (variants/lib_prof.so: -DZYG_PROFILE_SECTIONS on the POOL scenes' units) of the glass Cornell box and the outdoor scene
. "$(dirname "$0")/common.sh"
set +e
export ZYG_AMD_LIB=$R/variants/lib_prof.so
for scene in glass outdoor; do for precise in 1 0; do
  echo "== $scene precise=$precise"
  ZYG_SCENE=$scene ZYG_PRECISE=$precise timeout 600 python3 tools/sections.py substitute 2>&1 | grep -v "^Sky"
done; done
