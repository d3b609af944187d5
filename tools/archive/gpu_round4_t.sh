#!/bin/bash
# round 4: does parsing ahead of the window (slots=6, window=4, ahead=2) help the boundary on every call shape?  (in-process A/B + the multi-caller tool)
cd $GRAFT_REPO_ROOT
C="slots=6,window=4,ahead=2"
timeout 600 python tools/boundary_ab.py 65536 5 "-" "$C" "slots=6,window=4,ahead=1" 2>&1 | grep config
timeout 600 python tools/boundary_ab.py 8192 12 "-" "$C" 2>&1 | grep config
timeout 600 python tools/boundary_ab.py 4096 12 "-" "$C" 2>&1 | grep config
for t in "" "$C"; do MINA_TUNE=$t timeout 600 python tools/batch_callers.py 4 8192 12 2>/dev/null | tail -1; MINA_TUNE=$t timeout 600 python tools/batch_callers.py 2 8192 12 2>/dev/null | tail -1; done
