#!/bin/bash
# GPU box: parity tests, kernel time of every build present, then per-env dynamic instruction counts.
set -e
bash tools/ygo_variants.sh
bash tools/pmc_ygo_skip.sh
