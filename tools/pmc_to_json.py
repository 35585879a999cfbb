"""rocprofv3 --pmc counter_collection CSVs -> one JSON with the mean counter value per kernel and launch.

    python tools/pmc_to_json.py OUT.json DIR [DIR ...]      (each DIR = the -d directory of one rocprofv3 --pmc pass)

bench.py reads the newest profiles/rNN_pmc_kernels.json for roofline.traffic / mfma_util (Measurement)."""
import glob
import json
import sys

import pandas as pd


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    frames = []
    for d in dirs:
        prefix = ''
        if ':=' in d:                                   # "tts:=/tmp/dir": kernels of that pass are filed as "tts:<kernel>"
            prefix, d = d.split(':=', 1)
            prefix += ':'
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            fr = pd.read_csv(f)
            fr['_prefix'] = prefix
            frames.append(fr)
    if not frames:
        raise SystemExit('no counter_collection.csv under ' + ' '.join(dirs))
    t = pd.concat(frames)
    t = t[~t.Kernel_Name.str.contains('at::|rocclr|Cijk|vectorized_elementwise')]
    base = t.Kernel_Name.str.split('(').str[0].str.replace(r'^void ', '', regex=True)
    # template arguments are dropped, except for td16_layer_kernel: <false> is an encoder layer, <true> the last layer + the pooling tail
    t['k'] = t['_prefix'] + base.where(base.str.startswith('td16_layer_kernel'), base.str.replace(r'<.*', '', regex=True))
    g = t.groupby(['k', 'Counter_Name'])['Counter_Value'].mean()
    n = t.groupby(['k', 'Counter_Name'])['Counter_Value'].count()
    kernels = {}
    for (k, c), v in g.items():
        kernels.setdefault(k, {})[c] = float(v)
        kernels[k].setdefault('_launches_sampled', int(n[(k, c)]))
    for k, reg in t.groupby('k')[['VGPR_Count', 'LDS_Block_Size', 'Scratch_Size']].max().iterrows() if 'VGPR_Count' in t.columns else []:
        kernels[k]['_vgpr'] = int(reg['VGPR_Count']); kernels[k]['_lds'] = int(reg['LDS_Block_Size']); kernels[k]['_scratch'] = int(reg['Scratch_Size'])
    if 'Accum_VGPR_Count' in t.columns:
        for k, v in t.groupby('k')['Accum_VGPR_Count'].max().items():
            kernels[k]['_agpr'] = int(v)
    if 'Start_Timestamp' in t.columns and 'End_Timestamp' in t.columns:
        # launch duration UNDER the pass that sampled GRBM_GUI_ACTIVE: cycles / duration = the shader clock of that launch
        tt = t[t.Counter_Name == 'GRBM_GUI_ACTIVE']
        for k, v in ((tt.End_Timestamp - tt.Start_Timestamp) * 1e-6).groupby(tt.k).mean().items():
            kernels[k]['_ms'] = float(v)
            # GRBM_GUI_ACTIVE / 8 (it is summed over the XCDs) over the launch's duration is the shader clock only when the launch
            # is long against its dispatch / drain time: below ~50 us the counter also ticks before and after the timestamps and the
            # quotient comes out above the chip's 2.4 GHz (r04: 3.0-3.5 GHz for the td_* / pool_* kernels).  Not a clock there.
            if v >= 0.05:
                kernels[k]['_shader_clock_mhz'] = round(kernels[k]['GRBM_GUI_ACTIVE'] / 8.0 / (v * 1e-3) / 1e6, 1)
    with open(out, 'w') as f:
        json.dump({'units': 'mean per launch; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE counts half the bytes of wide reads on gfx950)',
                   'kernels': kernels}, f, indent=1, sort_keys=True)
    for k in sorted(kernels):
        print(k, {c: round(v, 1) for c, v in kernels[k].items()})


if __name__ == '__main__':
    main()
