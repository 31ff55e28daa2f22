#!/bin/bash
# build the product library only (fast iteration); non-zero exit on any compile error
cd "$(dirname "$0")/../ark-blst_amd/csrc" && make -s -j8 ../lib/libarkblst_amd.so 2>&1 | grep -E "error|Error" ; test ${PIPESTATUS[0]} -eq 0
