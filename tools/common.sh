# sourced by the tools/*.sh scripts that run on the GPU box (gpurun): strict mode, the repo root, a writable TMPDIR
set -eu
R="${GRAFT_REPO_ROOT:?run these scripts through gpurun, which sets GRAFT_REPO_ROOT}"
export TMPDIR=/tmp
cd "$R"
