#!/bin/bash
# Whole-frame lookahead inside the reference encoder: oracle/_ref/enc_hip (the reference's own sources
# + libx265_hip.so) encodes the same clip with its own lookahead search, with every frame cost estimate's search done
# by one x265_hip_lookahead_batch launch and compared call by call with the reference's own (X265_USE_HIP_LOOKAHEAD=1),
# with the reference's search not run at all (=2, the timing run), and again (=3, =4) with the pairs of a picture searched
# in one launch ahead of their use.  The bitstreams must be identical.
#   tests/tools/lookahead_demo.sh [width height frames [pool_threads]]      DROPIN_FADE=1: weighted references
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
W=${1:-416}; H=${2:-240}; N=${3:-12}; T=${4:-4}
cd "$R" || exit 1
python3 - "$W" "$H" "$N" <<'PY' || exit 1
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from x265_amd import synth
W, H, N = (int(v) for v in sys.argv[1:4])
seq = synth.Sequence(W, H, seed=7, n_patches=3)
with open("/tmp/lookdemo.yuv", "wb") as f:
    for i in range(N):
        y = seq.frame(i).astype(np.float64)
        if os.environ.get("DROPIN_FADE") and i >= N // 3:
            y = y * max(0.15, 1.0 - 0.05 * (i - N // 3)) + 0.5
        f.write(np.clip(y, 0, 255).astype(np.uint8).tobytes())
        f.write(bytes([128]) * (W * H // 2))
PY
run() { timeout -k 10 600 oracle/_ref/enc_hip /tmp/lookdemo.yuv "$W" "$H" "$N" /tmp 0 0 0 "$T" 1 2>&1 | grep --line-buffered -v "Assembly not supported"; }
echo "== the reference's own lookahead search ($T pool threads)"
run | tee /tmp/look_c.txt || exit 1
echo "== frame searches by x265_hip_lookahead_batch, compared with the reference's own"
X265_USE_HIP_LOOKAHEAD=1 run | tee /tmp/look_v.txt || exit 1
echo "== frame searches by x265_hip_lookahead_batch alone"
X265_USE_HIP_LOOKAHEAD=2 run | tee /tmp/look_t.txt || exit 1
echo "== the pairs of a picture batched ahead of their use, compared with the reference's own"
X265_USE_HIP_LOOKAHEAD=3 run | tee /tmp/look_bv.txt || exit 1
echo "== batched ahead, x265_hip_lookahead_batch alone"
X265_USE_HIP_LOOKAHEAD=4 run | tee /tmp/look_bt.txt || exit 1
a=$(grep -o "hash [0-9a-f]*" /tmp/look_c.txt); b=$(grep -o "hash [0-9a-f]*" /tmp/look_v.txt); c=$(grep -o "hash [0-9a-f]*" /tmp/look_t.txt)
d=$(grep -o "hash [0-9a-f]*" /tmp/look_bv.txt); e=$(grep -o "hash [0-9a-f]*" /tmp/look_bt.txt)
if [ -n "$a" ] && [ "$a" == "$b" ] && [ "$a" == "$c" ] && [ "$a" == "$d" ] && [ "$a" == "$e" ] && grep -q "answered from them, 0 differed" /tmp/look_v.txt &&
   grep -q "answered from them, 0 differed" /tmp/look_bv.txt; then
  echo "LOOKAHEAD DROP-IN OK: identical bitstreams ($a)"
else
  echo "LOOKAHEAD DROP-IN MISMATCH: '$a' vs '$b' vs '$c' vs '$d' vs '$e'"; exit 1
fi
