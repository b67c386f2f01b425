# builds the round-4 library (git tag of the round-4 verdict commit) beside the tree, for same-box A/B runs of the micro benches
set -e
rm -rf /tmp/oldlib && mkdir -p /tmp/oldlib/src
git -C "$(dirname "$0")/../.." archive cda3f1e deepavfusion_amd/csrc include | tar -x -C /tmp/oldlib/src
make -C /tmp/oldlib/src/deepavfusion_amd/csrc -j8 > /dev/null
mkdir -p "$(dirname "$0")/lib_r04" && cp /tmp/oldlib/src/deepavfusion_amd/libdavfusion_hip.so "$(dirname "$0")/lib_r04/"
