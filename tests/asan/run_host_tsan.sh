#!/bin/bash
# ThreadSanitizer on the host threads of librtamd: the fan-out of rt_render_multi (one host thread per rank, abi.cpp::render_fanout) on the
# test-only device stub's fake devices, driven by tests/asan/tsan_fanout.cpp from two caller threads at once.  (SURVEY s5 "race
# detection"; the device side has no race detector on this pool.  A C++ driver, not pytest: CPython under a preloaded libtsan does not finish.)
set -e
cd "$(dirname "$0")/../.."
P=rust-raytracer_amd
g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=thread -Iinclude -I$P/csrc -o tests/asan/tsan_fanout \
    tests/asan/tsan_fanout.cpp $P/csrc/abi.cpp $P/csrc/host/scene.cpp $P/csrc/host/flatten.cpp $P/csrc/host/accel.cpp $P/csrc/host/loader.cpp $P/csrc/host/obj.cpp \
    $P/csrc/host/png.cpp $P/csrc/host/schedule.cpp tests/asan/device_stub.cpp -lpthread
TSAN_OPTIONS="halt_on_error=0 exitcode=66" RTAMD_STUB_DEVICES=4 tests/asan/tsan_fanout tests/golden/scenes/scene_10.json
