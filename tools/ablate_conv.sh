#!/bin/bash
# ablation of the conv kernel phases on the 64->64 @256^2 layer (debug bits: 1 no A staging, 2 no B staging, 4 no MFMA, 8 no epilogue)
for d in 0 1 2 3 4 8 12 7 15; do
  echo -n "debug=$d: "; CDNET_CONV_DEBUG=$d python tools/bench_conv.py 16 2>&1 | grep "enc1_2" | head -1
done
