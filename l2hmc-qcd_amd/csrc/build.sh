#!/usr/bin/env bash
# Build libl2q.so for gfx950, in-tree (see Makefile; `build.sh --clean` rebuilds from source).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
if [ "${1:-}" = "--clean" ]; then make -C "$here" clean >/dev/null; fi
make -C "$here" -j"$(nproc)" all 2>&1 | grep -v "^make" || true
test -f "$here/../l2hmc/_lib/libl2q.so"
