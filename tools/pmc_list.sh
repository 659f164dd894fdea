#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>&1 | grep -iE "ICACHE|IFETCH|INST_CACHE|SQC_" | cut -c1-160 | sort -u | head -60
