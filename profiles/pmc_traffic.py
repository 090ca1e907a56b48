"""HBM traffic of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) over `bench.py --steps 1 --warmup 1 --no-cpu-baseline`:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python3 profiles/pmc_traffic.py OUT/fetch/fetch_counter_collection.csv OUT/write/write_counter_collection.csv

Counters are in KB and summed over the XCDs by rocprofv3.  gfx950 correction: FETCH_SIZE reports half the bytes of
16-B-per-lane streaming reads -> doubled.  Writes profiles/lstm_gemm_traffic.json, which bench.py reports as
roofline.traffic (bytes per launch, averaged over the launches of the kernel).

    python3 profiles/pmc_traffic.py FETCH.csv WRITE.csv persist_ persist_decode_traffic.json 1

does the same for other kernels (here: the persistent encoder and decoder of `bench.py --workload c2`, averaged over both as
bench.py's kernel class does; last argument = factor on FETCH_SIZE: their hand-off loads are 8 B per lane, no doubling)."""
import csv
import json
import os
import sys

KERNEL = 'gemm_kernel<1, 1>'


def per_kernel_average(path, counter):
    total, n = 0.0, 0
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if KERNEL in row.get('Kernel_Name', '') and row.get('Counter_Name') == counter:
                total += float(row['Counter_Value'])
                n += 1
    return (total / n if n else 0.0), n


def main():
    global KERNEL
    outname = 'lstm_gemm_traffic.json'
    fetch_scale = 2.0
    if len(sys.argv) > 4:
        KERNEL, outname = sys.argv[3], sys.argv[4]
    if len(sys.argv) > 5:
        fetch_scale = float(sys.argv[5])
    fetch_kb, n1 = per_kernel_average(sys.argv[1], 'FETCH_SIZE')
    write_kb, n2 = per_kernel_average(sys.argv[2], 'WRITE_SIZE')
    out = {'hbm_bytes_per_launch': (fetch_scale * fetch_kb + write_kb) * 1024.0, 'fetch_scale': fetch_scale, 'fetch_kb_raw_avg': fetch_kb, 'write_kb_raw_avg': write_kb,
           'launches': n1, 'kernel': KERNEL,
           # which build the passes were taken on (bench.py names both beside roofline.traffic: the figure is a committed constant)
           'round': int(os.environ.get('CASV_PROFILE_ROUND', '0')) or None, 'commit': os.environ.get('CASV_PROFILE_COMMIT'),
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1; average over all '
                   'dispatches of the kernel; FETCH x fetch_scale (2: the gfx950 correction for 16-B-per-lane reads); '
                   'WRITE_SIZE uncalibrated for 4-B-per-lane stores.' + (' First version of round 1 (round-robin tile order): fetch '
                   '127710 KB raw, 290 MB per launch.' if outname == 'lstm_gemm_traffic.json' else '')}
    assert n1 and n1 == n2, (n1, n2)
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, outname), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
