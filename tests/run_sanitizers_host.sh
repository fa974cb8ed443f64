#!/bin/bash
# Host code under the sanitizers, on the CPU (GPU sanitizers are not available on this pool):
#   1. the host team (qc_host_team.h: worker pool, landing watch, ring re-arm, deadline) driven by tests/host_team_test.cpp with a thread
#      standing in for the copy engine, under -fsanitize=thread and under -fsanitize=address,undefined;
#   2. (with --lib) the library's host-only entry points under AddressSanitizer, as tests/run_asan_host.sh always did.
# usage: tests/run_sanitizers_host.sh [--lib] [iterations]
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/quantumcollocation.jl_amd/csrc
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
LIB=0
if [ "${1:-}" = "--lib" ]; then LIB=1; shift; fi
IT=${1:-12}
CXX=${CXX:-g++}
# (qc_host_copy.cpp -- the scans, fills and streaming copies that touch the landing block -- is left uninstrumented in the thread pass:
#  those words are written by the stand-in engine on purpose; in the address pass everything is instrumented)
$CXX -O1 -g -std=c++17 -c "$C/qc_host_copy.cpp" -o "$W/copy_plain.o"
$CXX -O1 -g -std=c++17 -fsanitize=thread -fno-omit-frame-pointer -I"$C" "$R/tests/host_team_test.cpp" "$W/copy_plain.o" -o "$W/team_tsan" -lpthread
TSAN_OPTIONS="suppressions=$R/tests/tsan_host_team.supp halt_on_error=1 second_deadlock_stack=1" "$W/team_tsan" "$IT"
echo "thread sanitizer: clean"
$CXX -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I"$C" "$R/tests/host_team_test.cpp" "$C/qc_host_copy.cpp" -o "$W/team_asan" -lpthread
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 "$W/team_asan" "$IT"
echo "address + undefined-behaviour sanitizers: clean"
if [ $LIB = 1 ]; then bash "$R/tests/run_asan_host.sh"; fi
