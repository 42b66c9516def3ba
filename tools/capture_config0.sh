#!/bin/bash
# BASELINE.json configs[0] as stated, once, on the GPU box: the reference encoder (--preset medium, C primitives) encodes
# 64 pictures of the 1920x1080 synthetic sequence with EVERY motionEstimate call recorded (full resolution and lookahead),
# and every call is replayed through x265_hip_me_batch (tests/tools/replay_capture.py gpu).  Too large to commit: the
# verdict line goes to profiles/.  Run through gpurun from the repository root: bash tools/capture_config0.sh [pictures]
set -e -o pipefail
N=${1:-64}
D=/tmp/enc_config0
mkdir -p "$D"
cd "$(dirname "$0")/.."
python3 - "$D" "$N" <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd import synth
W, H, N = 1920, 1080, int(sys.argv[2])
seq = synth.Sequence(W, H, seed=0x265)
with open(sys.argv[1] + "/in.yuv", "wb") as f:
    for i in range(N):
        f.write(seq.frame(i).tobytes())
        f.write(bytes([128]) * (W * H // 2))
PY
echo "capturing $N pictures" 
X265_CAPTURE_SPARSE=1 timeout -k 10 700 oracle/_ref/enc_capture "$D/in.yuv" 1920 1080 "$N" "$D" 100000000 100000000 0 16 1 2> "$D/capture.log" || { tail -n 20 "$D/capture.log"; exit 1; }
tail -n 3 "$D/capture.log"
rm -f "$D/in.yuv"
ls -la "$D"
timeout -k 10 600 python3 -u tests/tools/replay_capture.py "$D" gpu
