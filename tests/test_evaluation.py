"""nisqa_amd/evaluation.py against the reference's eval_results (golden vectors from
tests/golden/make_golden_eval.py, and the live reference where /root/reference exists)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import helpers  # noqa: F401  (sys.path)
from nisqa_amd import evaluation as ev, synth
from nisqa_amd import NISQA_lib as NL

HERE = os.path.dirname(os.path.abspath(__file__))
MAPPINGS = [None, 'first_order', 'second_order', 'third_order_not_monotonic', 'third_order']


def _golden():
    with open(os.path.join(HERE, 'golden', 'eval_reference.json')) as f:
        return json.load(f)


def _close(a, b, tol):
    a = np.array([np.nan if v is None else v for v in a], dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(b[~np.isnan(a)], a[~np.isnan(a)], rtol=0, atol=tol)


@pytest.mark.parametrize('with_con', [True, False])
@pytest.mark.parametrize('mapping', MAPPINGS)
def test_eval_results_matches_reference_fixture(mapping, with_con):
    g = _golden()['%s|%s' % ('con' if with_con else 'file', mapping)]
    df, dcon = synth.eval_corpus(11)
    res, overall = NL.eval_results(df, dcon=dcon if with_con else None, target_mos='mos', target_ci='mos_ci',
                                   pred='mos_pred', mapping=mapping)
    tol = 2e-5 if mapping == 'third_order' else 1e-9                # SLSQP: same optimiser, same start, float noise
    want = pd.DataFrame(g['db_results']['data'], columns=g['db_results']['columns'])
    assert list(res.columns) == list(want.columns) and list(res['db']) == list(want['db'])
    for c in want.columns[1:]:
        _close(list(want[c]), res[c].to_numpy(), tol)
    assert list(overall) == list(g['overall'])
    _close(list(g['overall'].values()), list(overall.values()), tol)
    _close(g['y_hat_map'], df['y_hat_map'].to_numpy(), tol)


def test_eval_results_against_live_reference():
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip('reference tree not on this machine')
    RL = ref_shim.import_reference_lib()
    df, _ = synth.eval_corpus(23, n_db=4, n_con=9, per_con=5)
    df2 = df.copy()
    for mapping in (None, 'first_order', 'second_order'):
        # per-file block only: the reference's per-condition block does not run under pandas 2 (string columns)
        want, ow = RL.eval_results(df2, dcon=None, mapping=mapping)
        got, og = ev.eval_results(df, dcon=None, mapping=mapping)
        pd.testing.assert_frame_equal(got, want, check_exact=False, rtol=0, atol=1e-10)
        assert list(og) == list(ow)
        np.testing.assert_allclose(np.array(list(og.values()), float), np.array(list(ow.values()), float), atol=1e-10)


def test_metrics_definitions_p1401():
    y = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    yh = np.array([1.5, 1.5, 3.0, 4.5, 4.0])
    assert ev.calc_rmse(y, yh) == pytest.approx(np.sqrt(np.mean((y - yh) ** 2)))
    assert ev.calc_rmse(y, yh, d=1) == pytest.approx(np.sqrt(np.sum((y - yh) ** 2) / 4))
    assert np.isnan(ev.calc_rmse(y[:1], yh[:1], d=1))
    ci = np.array([0.6, 0.2, 0.1, 0.1, 0.5])
    star, p_err, err = ev.calc_rmse_star(y, yh, ci, 1)
    np.testing.assert_allclose(p_err, [0, 0.3, 0, 0.4, 0.5])
    assert star == pytest.approx(np.sqrt((0.09 + 0.16 + 0.25) / 4))
    assert np.isnan(ev.calc_rmse_star(y, yh, np.array([np.nan] * 5), 1)[0])
    np.testing.assert_allclose(ev.calc_mapped(yh, np.array([1.0, 2.0, 0.5])), 1 + 2 * yh + 0.5 * yh ** 2)
    b = ev.fit_first_order(3 * yh - 1, yh)
    np.testing.assert_allclose(b, [-1, 3], atol=1e-12)
    r = ev.calc_eval_metrics(y, np.full(5, 2.0))
    assert np.isnan(r['r_p']) and r['rmse'] == pytest.approx(np.sqrt(np.mean((y - 2) ** 2)))
    assert ev.is_const(np.full(4, 3.3)) and not ev.is_const(yh)
    b3 = ev.fit_monotonic_third_order(pd.DataFrame({'mos': y, 'mos_pred': yh}), pred='mos_pred', target_mos='mos',
                                      target_ci='mos_ci', mapping='error')
    grid = np.arange(1.49, 4.51, 0.1)
    assert (b3[1] + 2 * b3[2] * grid + 3 * b3[3] * grid ** 2 > -1e-6).all()
    with pytest.raises(NotImplementedError):
        ev.calc_mapping(pd.DataFrame({'mos': y, 'mos_pred': yh}), mapping='fourth_order', target_mos='mos', pred='mos_pred')


def test_unlabelled_database_and_string_columns():
    df, dcon = synth.eval_corpus(5)
    df.loc[df.db == 'DB_A', 'mos'] = np.nan                          # a database without subjective scores
    res, overall = ev.eval_results(df, dcon=dcon, mapping='first_order')
    a = res[res.db == 'DB_A'].iloc[0]
    assert np.isnan(a['r_p_file']) and np.isnan(a['rmse_map_file']) and np.isnan(a['r_p_con'])
    b = res[res.db == 'DB_B'].iloc[0]
    assert 0.5 < b['r_p_file'] <= 1 and 0.5 < b['r_p_con'] <= 1 and b['rmse_star_map_con'] <= b['rmse_map_con']
    assert np.isfinite(overall['r_p_mean_con']) and np.isnan(overall['r_p_all'])
    assert df.loc[df.db == 'DB_A', 'y_hat_map'].isna().all() and df.loc[df.db == 'DB_B', 'y_hat_map'].notna().all()
