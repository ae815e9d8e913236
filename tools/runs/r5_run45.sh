#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for g in owner bucket owner; do BAND_GRADIENT=$g BAND_LAYOUTS=bands python tools/band_timing.py 1 cfg2 2>/dev/null | cut -c1-400; done
