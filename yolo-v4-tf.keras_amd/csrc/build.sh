#!/bin/bash
# Builds libyolo4hip.so in-tree for gfx950 (see build.py: content-hash incremental build; Y4_CLEAN=1 forces a clean one).
set -e
exec python3 "$(dirname "$0")/build.py" "$@"
