#!/bin/bash
# Drop-in demonstration: the reference encoder (oracle/_ref/enc_hip = the reference's own sources
# + source/common/hip/hip-primitives.cpp + libx265_hip.so) encodes the same clip five times: with its C primitives,
# with the HIP primitive table installed, with every MotionEstimate::motionEstimate call (the lookahead's included)
# answered by x265_hip_me_batch, and with the luma block of every TComPrediction::motionCompensation call answered
# by x265_hip_predict_batch, and with the lookahead planes (Lowres::init) and weighted reference planes
# (MotionReference::applyWeight) made by x265_hip_lowres_init / x265_hip_plane_weight; the bitstreams must be
# identical and no answer may differ from the reference's.
# DROPIN_SKIP_TABLE=1 (large clips) skips the per-call table and motionCompensation runs; DROPIN_FADE=1 fades the
# clip to black so that the encoder's weight analysis picks non-unit weights.
#   tests/tools/dropin_demo.sh [width height frames]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
W=${1:-416}; H=${2:-240}; N=${3:-6}
cd "$R" || exit 1
python3 - "$W" "$H" "$N" <<'PY' || exit 1
import sys
import numpy as np
sys.path.insert(0, ".")
from x265_amd import synth
W, H, N = (int(v) for v in sys.argv[1:4])
seq = synth.Sequence(W, H, seed=7, n_patches=3)
with open("/tmp/dropin.yuv", "wb") as f:
    import os
    for i in range(N):
        y = seq.frame(i).astype(np.float64)
        if os.environ.get("DROPIN_FADE") and i >= N // 3:   # fade to black over the last two thirds: weighted references
            y = y * max(0.15, 1.0 - 0.05 * (i - N // 3)) + 0.5
        f.write(np.clip(y, 0, 255).astype(np.uint8).tobytes())
        f.write(bytes([128]) * (W * H // 2))
PY
echo "== C primitives"
timeout -k 10 300 oracle/_ref/enc_hip /tmp/dropin.yuv "$W" "$H" "$N" /tmp 0 0 0 4 1 2>&1 | grep -v "Assembly not supported" | tee /tmp/dropin_c.txt || exit 1
echo "== HIP primitive table"
if [ -n "$DROPIN_SKIP_TABLE" ]; then   # a 1080p clip makes tens of millions of primitive calls at ~20 us each
  echo "(skipped)"; sed 's/^enc_capture: [0-9]* pictures/& [copied from the C run]/' /tmp/dropin_c.txt > /tmp/dropin_hip.txt
  echo "enc_capture: HIP primitive table installed (skipped)" >> /tmp/dropin_hip.txt
else
X265_USE_HIP=1 timeout -k 10 900 oracle/_ref/enc_hip /tmp/dropin.yuv "$W" "$H" "$N" /tmp 0 0 0 4 1 2>&1 | grep -v "Assembly not supported" | tee /tmp/dropin_hip.txt || exit 1
fi
echo "== motionEstimate answered by x265_hip_me_batch (C primitives elsewhere)"
X265_USE_HIP_ME=1 timeout -k 10 900 oracle/_ref/enc_hip /tmp/dropin.yuv "$W" "$H" "$N" /tmp 0 0 0 4 1 2>&1 | grep --line-buffered -v "Assembly not supported" | tee /tmp/dropin_me.txt || exit 1
echo "== motionCompensation (luma) answered by x265_hip_predict_batch (C primitives elsewhere)"
if [ -n "$DROPIN_SKIP_TABLE" ]; then   # one plane download per call: small clips only
  echo "(skipped)"; d_skip=1
else
X265_USE_HIP_MC=1 timeout -k 10 900 oracle/_ref/enc_hip /tmp/dropin.yuv "$W" "$H" "$N" /tmp 0 0 0 4 1 2>&1 | grep --line-buffered -v "Assembly not supported" | tee /tmp/dropin_mc.txt || exit 1
d=$(grep -o "hash [0-9a-f]*" /tmp/dropin_mc.txt)
grep -q "predicted on the GPU.* 0 differed" /tmp/dropin_mc.txt || { echo "DROP-IN MISMATCH in motionCompensation"; exit 1; }
fi
echo "== lookahead planes and weighted reference planes made by x265_hip_lowres_init / x265_hip_plane_weight"
X265_USE_HIP_PLANES=1 timeout -k 10 900 oracle/_ref/enc_hip /tmp/dropin.yuv "$W" "$H" "$N" /tmp 0 0 0 4 1 2>&1 | grep --line-buffered -v "Assembly not supported" | tee /tmp/dropin_pl.txt || exit 1
e=$(grep -o "hash [0-9a-f]*" /tmp/dropin_pl.txt)
grep -q "lookahead planes made on the GPU, 0 differed; [0-9]* weighted reference planes made on the GPU, 0 differed" /tmp/dropin_pl.txt || { echo "DROP-IN MISMATCH in plane production"; exit 1; }
a=$(grep -o "hash [0-9a-f]*" /tmp/dropin_c.txt); [ -n "$d_skip" ] && d=$a; b=$(grep -o "hash [0-9a-f]*" /tmp/dropin_hip.txt); c=$(grep -o "hash [0-9a-f]*" /tmp/dropin_me.txt)
if [ -n "$a" ] && [ "$a" == "$b" ] && [ "$a" == "$c" ] && [ "$a" == "$d" ] && [ "$a" == "$e" ] && grep -q "HIP primitive table installed" /tmp/dropin_hip.txt &&
   [ "$(grep -c "answered on the GPU, 0 differed" /tmp/dropin_me.txt)" == 2 ]; then   # full-resolution and lookahead calls
  echo "DROP-IN OK: identical bitstreams ($a)"
else
  echo "DROP-IN MISMATCH: '$a' vs '$b' vs '$c'"; exit 1
fi
