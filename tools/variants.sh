#!/bin/bash
# time the fused kernel variants on the config-2 shape (GPU box)
for sc in poly hw hwraw; do for k in mfma valu; do AOG_SINCOS=$sc python tools/step_loop.py --kernel $k "$@" | sed "s/^/sincos=$sc /"; done; done
