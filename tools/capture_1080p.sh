#!/bin/bash
# Make the 1080p medium-preset capture of the reference encoder on this machine (first 130,000 full-resolution
# motionEstimate calls of 34 pictures, as profiles/r01_final_encoder_calls_timing.txt) into $1 (default /tmp/enc1080).
set -e
D=${1:-/tmp/enc1080}
mkdir -p "$D"
cd "$(dirname "$0")/.."
python3 - "$D" <<'PY'
import sys
sys.path.insert(0, ".")
from x265_amd import synth
W, H, N = 1920, 1080, 34
seq = synth.Sequence(W, H, seed=0x265)
with open(sys.argv[1] + "/in.yuv", "wb") as f:
    for i in range(N):
        f.write(seq.frame(i).tobytes())
        f.write(bytes([128]) * (W * H // 2))
PY
oracle/_ref/enc_capture "$D/in.yuv" 1920 1080 34 "$D" 130000 0 0 16 1 2> "$D/capture.log" || { tail -n 20 "$D/capture.log"; exit 1; }
tail -n 3 "$D/capture.log"
rm -f "$D/in.yuv"
