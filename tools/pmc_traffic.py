"""Aggregate rocprofv3 PMC passes into HBM bytes per launch per kernel -> profiles/pmc_traffic_rNN.json (bench.py reads it for
`roofline.traffic`).

    # on the GPU box, two SEPARATE passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --encode-only --no-graph --groups 1 --steps 3
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --encode-only --no-graph --groups 1 --steps 3
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic_rNN.json

Counter values are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at
64 B -> doubled for the wide coalesced streams these kernels issue; WRITE_SIZE is taken as is (r01 calibration: it matched the
C matrix of the GEMM exactly)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*', '', name).replace('void ', '')
    return name.replace('pk::gemm', 'gemm').replace(' ', '') if name.startswith('pk::gemm') else name.replace(' ', '')


def collect(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter:
                    continue
                a = agg[short(row['Kernel_Name'])]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    return agg


def table(fetch_dir, write_dir):
    fetch, write = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not (name.startswith('pk::') or name.startswith('gemm')):
            continue
        nf, kf = fetch.get(name, [0, 0.0])
        nw, kw = write.get(name, [0, 0.0])
        rd = 2.0 * 1024.0 * kf / nf if nf else 0.0
        wr = 1024.0 * kw / nw if nw else 0.0
        kernels[name] = dict(dispatches=max(nf, nw), fetch_bytes_per_launch_x2=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                             note='fabric-side requests of the L2 (TCC_EA0): Infinity-Cache hits are included')
    return kernels


def main():
    """pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json [SECTION FETCH_DIR WRITE_DIR ...]: the first pair is the bf16 encode leg ('kernels', what
    bench.py's headline roofline reads); further triples add per-leg sections ('sample', 'bf16x3_encode', 'bf16x3_sample': the same kernel
    template runs other shapes there, so their bytes per launch are kept apart)."""
    fetch_dir, write_dir, out = sys.argv[1:4]
    kernels = table(fetch_dir, write_dir)
    src = ('separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --encode-only --no-graph` (sections: of the named leg), '
           'mean per dispatch, FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B), tools/pmc_traffic.py')
    doc = dict(source=src, kernels=kernels, sections={})
    rest = sys.argv[4:]
    for i in range(0, len(rest) - 2, 3):
        doc['sections'][rest[i]] = table(rest[i + 1], rest[i + 2])
    json.dump(doc, open(out, 'w'), indent=1)
    for sec, tab in [('encode', kernels)] + list(doc['sections'].items()):
        print(f'-- {sec}')
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'])[:14]:
            print(f"{k[:90]:90s} n={v['dispatches']:5d} read {v['fetch_bytes_per_launch_x2'] / 1e6:8.2f} MB write {v['write_bytes_per_launch'] / 1e6:8.2f} MB")


if __name__ == '__main__':
    main()
