#!/bin/bash
# Run ON THE GPU BOX: the executable on a genbase dump with RFI, once with the channeliser that flags its own rows
# (default) and once with the kurtosis kernel + channeliser pair (PB_FUSE_KURTOSIS=0), both hosts: the four pairs of
# .fil files must be byte-identical.  Exercises real staging (H2D per second into reused buffer sets) at full size.
# usage: tools/soak_equivalence.sh [seconds of data = 8]
set -e
T=${1:-8}
D=/tmp/soak_eq; rm -rf $D; mkdir -p $D
python -m vlite-fast_amd.genbase -t $T -r 7 -d 2 -p 0.05 -f --out $D/dump.vdif > $D/genbase.log 2>&1
run() {  # name, env assignment, host command...
    local name=$1 envs=$2; shift 2
    mkdir -p $D/$name
    env $envs "$@" --replay $D/dump.vdif -w 2 -b 8 -P 1 -r 2 -o --datadir $D/$name --logdir $D/$name --no-control > $D/$name/log 2>&1
}
run nat_fused PB_FUSE_KURTOSIS=1 vlite-fast_amd/csrc/process_baseband
run nat_two   PB_FUSE_KURTOSIS=0 vlite-fast_amd/csrc/process_baseband
run py_fused  PB_FUSE_KURTOSIS=1 python -m vlite-fast_amd.process_baseband
run py_two    PB_FUSE_KURTOSIS=0 python -m vlite-fast_amd.process_baseband
ok=1
for f in $D/nat_fused/*.fil; do
    b=$(basename $f)
    for other in nat_two py_fused py_two; do
        g=$(ls $D/$other/*${b#*_} 2>/dev/null | head -1)
        [ -n "$g" ] || g=$D/$other/$b
        cmp "$f" "$g" || ok=0
    done
    ls -la "$f"
done
# round 4: the coadder host on this ONE antenna gives a coadded file whose payload is the antenna's excised stream
# (scale 1 / sqrt 1)
mkdir -p $D/co
python vlite-fast_amd/coadd_host.py --replay $D/dump.vdif -w 0 -b 8 -r 2 --datadir $D/co --logdir $D/co > $D/co/log 2>&1
python3 - $D <<'PY' || ok=0
import glob, importlib, sys
sys.path.insert(0, ".")
sp = importlib.import_module("vlite-fast_amd.sigproc")
d = sys.argv[1]
kur = open(glob.glob(d + "/nat_fused/*_kur.fil")[0], "rb").read()
co = open(glob.glob(d + "/co/*_ea99_kur.fil")[0], "rb").read()
hk, nk = sp.read_header(kur)
hc, nc = sp.read_header(co)
assert hc["telescope_id"] == 99 and {k: v for k, v in hk.items() if k != "telescope_id"} == {k: v for k, v in hc.items() if k != "telescope_id"}
assert kur[nk:] == co[nc:], "coadd of one antenna differs from its excised stream"
print("coadd_host on one antenna: payload = the antenna's _kur.fil (%d bytes)" % (len(co) - nc))
PY
[ $ok = 1 ] && echo "soak_equivalence: $T s, all .fil files identical" || { echo "soak_equivalence: MISMATCH"; exit 1; }
