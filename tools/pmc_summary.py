#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs) per kernel family and
write profiles/pmc_traffic.json.  Corrections (MI355X_MICROARCH.md "HBM"; verified here on a device copy of known size,
see profiles/r1_pmc_headline.txt): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a
wide coalesced streaming read -> x2; WRITE_SIZE needs no correction for 16 B/lane streaming stores."""
import collections
import csv
import json
import sys

csv.field_size_limit(10 ** 9)


def family(n):
    """Kernel name -> the key bench.py uses (ops.KERNEL_IDS)."""
    if 'act_apply_kernel<true' in n:
        return 'apply_online'
    if 'act_apply_kernel<false' in n:
        return 'apply_offline'
    if 'absmax_per_sample' in n:
        return 'stat'
    if 'histogram_kernel' in n:
        return 'histogram'
    if 'dwconv3x3' in n or 'pwdw_kernel' in n:       # (round 6: the fused launch of a recompute pair is accounted as its depthwise layer)
        return 'dwconv'
    if 'pw_stat_kernel' in n:                        # ... and the statistic-only pass as its pointwise layer
        return 'pwconv'
    if 'stem_conv3x3s2_kernel' in n or 'stem_mfma_kernel' in n or 'stem7_pool_kernel' in n or 'stem7_pool_lds_kernel' in n or 'stem3_rows_kernel' in n:
        return 'stem'
    if 'conv3x3_i8_kernel' in n:
        return 'conv3x3'
    if 'bn_act_stat_kernel' in n or 'bn_act_maxpool_stat_kernel' in n or 'add_act_stat_kernel' in n:
        return 'bn_act'
    if 'gap_stat_kernel' in n or 'gap_stat_lds_kernel' in n:
        return 'pool'
    if 'minmax_kernel' in n:
        return 'global_max'
    if 'pwconv_rows_kernel' in n:
        return 'dense'                       # the classifier on the codes: its own family (bench.py: FQ_KERNEL_DENSE)
    if 'pwconv_' in n or 'quant_transpose_i8_kernel' in n:
        return 'pwconv'
    if 'weight_codes_kernel' in n or 'weight_rows_lds_kernel' in n or 'weight_apply_kernel' in n:
        return 'weight'
    if 'direct_copy' in n or 'copyBuffer' in n:
        return 'device_copy(calibration)'
    return None


def load(path, cname):
    """family -> (counter values, number of library calls).  One fq_pwconv_i8 call launches two kernels (quantise +
    transpose, then the int8 GEMM) that the library's event scope brackets together, so its traffic is summed per CALL."""
    agg, calls = collections.defaultdict(list), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != cname:
            continue
        k = family(r['Kernel_Name'])
        if k:
            agg[k].append(float(r['Counter_Value']))
            if not (k == 'pwconv' and 'quant_transpose' in r['Kernel_Name']):
                calls[k] += 1
    return agg, calls


def main(fetch_csv, write_csv, out_json=None, tag=""):
    (f, fc), (w, wc) = load(fetch_csv, 'FETCH_SIZE'), load(write_csv, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        nf, nw = fc.get(k, 0), wc.get(k, 0)
        rd = 2.0 * 1024.0 * sum(fv) / max(nf, 1)
        wr = 1024.0 * sum(wv) / max(nw, 1)
        res[k] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                  "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
        print("%-28s calls %4d/%4d  read %10.1f MB  write %10.1f MB  total %10.1f MB per call"
              % (k, nf, nw, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
    if out_json:
        json.dump({"source": tag, "kernels": res}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, sys.argv[4] if len(sys.argv) > 4 else "")
