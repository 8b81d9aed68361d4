# sourced by the tools/*.sh scripts that run on the GPU box (gpurun): strict mode, the repo root, a writable TMPDIR
set -eu
R="${GRAFT_REPO_ROOT:?run these scripts through gpurun (GRAFT_REPO_ROOT is the snapshot's root)}"
export TMPDIR=/tmp
cd "$R"
