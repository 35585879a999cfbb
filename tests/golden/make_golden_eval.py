"""Golden vectors for nisqa_amd/evaluation.py, produced by the REFERENCE's own eval_results
(/root/reference/nisqa/NISQA_lib.py:1687-1852) on a seeded synthetic corpus.

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py
DEVIATION needed to run the reference here: its per-condition block calls ``df.groupby('con').mean()`` on frames
that carry string columns, which pandas >= 2 refuses; pandas 1 (the reference's env.yml pins 1.1) dropped such
columns silently.  That behaviour is restored for the duration of the call by defaulting numeric_only=True.
"""
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_shim                                    # noqa: E402
from nisqa_amd import synth                                    # noqa: E402

MAPPINGS = [None, 'first_order', 'second_order', 'third_order_not_monotonic', 'third_order']


def main():
    NL = ref_shim.import_reference_lib()
    from pandas.core.groupby.generic import DataFrameGroupBy
    orig = DataFrameGroupBy.mean
    DataFrameGroupBy.mean = lambda self, numeric_only=True, **kw: orig(self, numeric_only=numeric_only, **kw)
    out = {}
    try:
        for with_con in (True, False):
            for mapping in MAPPINGS:
                df, dcon = synth.eval_corpus(11)
                res, overall = NL.eval_results(df, dcon=dcon if with_con else None, target_mos='mos', target_ci='mos_ci',
                                               pred='mos_pred', mapping=mapping, do_print=False, do_plot=False)
                key = '%s|%s' % ('con' if with_con else 'file', mapping)
                out[key] = {'db_results': json.loads(res.to_json(orient='split')),
                            'overall': {k: (None if np.isnan(v) else float(v)) for k, v in overall.items()},
                            'y_hat_map': [None if np.isnan(v) else float(v) for v in df['y_hat_map']]}
    finally:
        DataFrameGroupBy.mean = orig
    with open(os.path.join(HERE, 'eval_reference.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('wrote', len(out), 'cases')


if __name__ == '__main__':
    main()
