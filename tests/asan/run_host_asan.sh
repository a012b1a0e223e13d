#!/bin/bash
# Host-side sanitizer run: builds the host half of librtamd + the oracle with ASan/UBSan and runs the CPU tests on them.
set -e
cd "$(dirname "$0")/../.."
P=rust-raytracer_amd
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -Iinclude -I$P/csrc -shared -o tests/asan/librtamd_host_asan.so \
    $P/csrc/abi.cpp $P/csrc/host/scene.cpp $P/csrc/host/flatten.cpp $P/csrc/host/accel.cpp $P/csrc/host/loader.cpp $P/csrc/host/obj.cpp \
    $P/csrc/host/png.cpp $P/csrc/host/schedule.cpp tests/asan/device_stub.cpp
make -s -C oracle librt_oracle_asan.so
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 RTAMD_HIP_RUNTIME=system RTAMD_LIB=$PWD/tests/asan/librtamd_host_asan.so \
  ORACLE_LIB=$PWD/oracle/librt_oracle_asan.so python -m pytest tests/test_host_loader.py tests/test_abi_symbols.py tests/test_schedule.py tests/test_oracle_kat.py \
  tests/test_oracle_vec3.py tests/test_golden.py tests/test_mixture.py tests/test_sppm.py -q -m "not gpu" -p no:cacheprovider "$@"
# the fan-out / partition / gather / stitch logic of rt_render_multi on the stub's fake devices (host memory; see device_stub.cpp)
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 RTAMD_HIP_RUNTIME=system RTAMD_LIB=$PWD/tests/asan/librtamd_host_asan.so RTAMD_STUB_DEVICES=4 \
  python -m pytest tests/test_multi_stub.py -q -p no:cacheprovider "$@"
