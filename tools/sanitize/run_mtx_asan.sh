#!/bin/bash
# AddressSanitizer + UBSan over the host-side .mtx parser (CPU build only; GPU sanitizers are not
# available on the pool).  Builds csrc/mtx.cpp with g++ and feeds it well-formed and malformed files.
set -eu
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
work="$(mktemp -d)"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ \
    -I/opt/rocm/include -I"$root/cellcomm_amd/csrc" "$here/mtx_harness.cpp" "$root/cellcomm_amd/csrc/mtx.cpp" -o "$work/harness"
python3 - "$work" <<'PY'
import sys, numpy as np
w = sys.argv[1]
rng = np.random.default_rng(1)
cases = {'empty': '', 'hdr_only': 'a\nb\nc\n', 'two_lines': 'a\nb', 'bad1': 'a\nb\nc\n1 2\n', 'bad2': 'a\nb\nc\n1 2 x\n',
         'bad3': 'a\nb\nc\n1.5 2 3\n', 'nonl': 'a\nb\nc\n1 2 3', 'crlf': 'a\r\nb\r\nc\r\n1\t2\t3\r\n\r\n4 5 6.5e0\r\n',
         'neg': 'a\nb\nc\n-5 -7 -1.25\n-5 -7 2\n9223372036854775 3 1\n', 'long': 'a\nb\nc\n' + '1' * 40 + ' 2 3\n',
         'spaces': 'a\nb\nc\n   \n  1   2   3   \n'}
for k, v in cases.items():
    open(f'{w}/{k}.mtx', 'w').write(v)
with open(f'{w}/rand.mtx', 'w') as f:
    f.write('h\nh\nh\n')
    for _ in range(20000):
        f.write(f'{rng.integers(-100, 5000)} {rng.integers(0, 3000)} {rng.integers(0, 99)}\n')
PY
"$work/harness" "$work"/*.mtx "$root/tests/golden/example_matrix.mtx"
echo "ASan/UBSan: clean"
rm -rf "$work"
