#!/bin/bash
# build libtnr_hip.so; print the tail of the log and fail loudly on error
cd "$(dirname "$0")/../tiny-newsrec_amd/csrc" || exit 1
if make -j8 > /tmp/tnr_make.log 2>&1; then echo "BUILD OK"; grep -E "warning" /tmp/tnr_make.log | head -5; python3 ../../tools/check_pending_spill.py || exit 1; else echo "BUILD FAILED"; grep -E "error" -A3 /tmp/tnr_make.log | head -30; exit 1; fi
