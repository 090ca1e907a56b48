"""HBM-side traffic of a WHOLE bench step (all kernels) from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes), for the workloads whose roofline line is not one kernel's (c4: all GEMMs of
the train step):

    python3 profiles/pmc_traffic_step.py FETCH.csv WRITE.csv STEPS OUT.json [kernel-name substrings whose reads are 16 B per lane ...]

Counters are in KB, summed over the XCDs by rocprofv3.  gfx950 correction: FETCH_SIZE reports half the bytes of 16-B-per-lane
streaming reads -- doubled for the kernels named on the command line (the GEMM kernels: all their tile loads are 16 B per lane),
taken as read for the others (uncalibrated widths).  Writes OUT.json with bytes per step, per kernel and in all."""
import csv
import json
import os
import sys


def per_kernel(path, counter):
    out = {}
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') == counter:
                name = row.get('Kernel_Name', '')
                t = out.setdefault(name, [0.0, 0])
                t[0] += float(row['Counter_Value']); t[1] += 1
    return out


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    steps, outname, wide = int(sys.argv[3]), sys.argv[4], sys.argv[5:]
    kernels, total = {}, 0.0
    for name in sorted(set(fetch) | set(write)):
        # (gemm_tn_split256_kernel stages its K-major operands with 4-byte loads: as read, like every other narrow reader)
        scale = 2.0 if any(w in name for w in wide) and 'gemm_tn_split' not in name else 1.0
        f, w = fetch.get(name, [0.0, 0]), write.get(name, [0.0, 0])
        by = (scale * f[0] + w[0]) * 1024.0 / steps
        total += by
        if by > 0.002 * 1e9:
            kernels[name[:90]] = {'bytes_per_step': by, 'fetch_scale': scale, 'dispatches_per_step': max(f[1], w[1]) / steps}
    out = {'hbm_bytes_per_launch': total, 'unit_of_launch': 'one bench step (all kernels)', 'steps_profiled': steps, 'kernels': kernels,
           'round': int(os.environ.get('CASV_PROFILE_ROUND', '0')) or None, 'commit': os.environ.get('CASV_PROFILE_COMMIT'),
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); FETCH doubled for kernels whose reads are 16 B per lane '
                   '(the gfx950 correction), other kernels as read',
           # counter collection serialises kernel dispatch: launches that must be resident TOGETHER take their one-launch form under it
           # (the train step's attention-cell backward: csrc/train.hip says so on stderr) -- the bytes are that form's
           'launch_forms': 'as under serialised dispatch (ROCPROF_COUNTERS set): the attention cell backward of the train step as ONE launch, '
                           'not the two co-resident launches of an unprofiled run'}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), outname), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}))


if __name__ == '__main__':
    main()
