#!/bin/bash
# run every tools/attnq64_x_* variant on the three judged shapes, print time + stamps (and correctness line for the non-drop ones)
for v in $(ls tools | grep attnq64_x_ | sed s/attnq64_//); do echo "=== $v"; for c in 9 11 12; do timeout 120 tools/attnq64_$v $c 2>&1 | grep -E "^----|q64 ksplit|stamps"; done; done
