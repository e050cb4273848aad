#!/bin/bash
# Build container only: extracts the c_inference_hip class from INTEGRATION.md section 2 and syntax-checks it against the
# reference's own headers (they need neither <mkl.h> nor rapidjson).  Nothing is linked or run.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=/root/reference/medgpc/src
[ -d "$REF" ] || { echo "reference tree not present: skipped"; exit 0; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
python3 - "$ROOT/INTEGRATION.md" "$TMP/c_inference_hip.h" <<'PY'
import re, sys
md = open(sys.argv[1]).read()
blocks = re.findall(r"```cpp\n(.*?)```", md, flags=re.S)
cls = [b for b in blocks if "class c_inference_hip" in b]
assert len(cls) == 1, "expected exactly one c_inference_hip snippet"
open(sys.argv[2], "w").write(cls[0])
PY
cat > "$TMP/tu.cpp" <<'CPP'
#include <cmath>
#include <cstdint>
#include <iostream>
#include <vector>
using namespace std;
#include "kernel/c_kernel.h"
#include "mean/c_meanfunc.h"
#include "likelihoods/c_likelihood.h"
#include "prior/c_prior.h"
#include "c_inference_hip.h"
int main() { c_inference_hip inf(1); inf.print_inffunc(); return 0; }
CPP
g++ -std=c++11 -fsyntax-only -w -I"$REF" -I"$ROOT/include" -I"$TMP" "$TMP/tu.cpp"
echo "INTEGRATION.md snippet: syntax OK against $REF headers"
