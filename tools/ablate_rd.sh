#!/bin/bash
# Variant builds of the 2-D chain kernels (CPU box) -> variants/abl/*.so; on the GPU box: tools/ablate_rd_run.sh
set -e
cd /root/repo; rm -rf variants/abl; mkdir -p variants/abl
B="bash tools/build_variant_rd.sh"
$B variants/abl/base.so
$B variants/abl/c9_32.so -DRSP_DOPPLER_COLS9=32
$B variants/abl/c9_16_xcd.so -DRSP_DOPPLER_XCDMAP=1
$B variants/abl/c9_32_xcd.so -DRSP_DOPPLER_COLS9=32 -DRSP_DOPPLER_XCDMAP=1
$B variants/abl/c10_16.so -DRSP_DOPPLER_COLS10=16
$B variants/abl/c10_16_xcd.so -DRSP_DOPPLER_COLS10=16 -DRSP_DOPPLER_XCDMAP=1
$B variants/abl/c10_8_plain.so -DRSP_DOPPLER_XCDMAP=0
