"""Corpus-level conformance statistics on the predictions of the HIP path.

Host-side mirror of the reference's evaluation helpers (nisqa/NISQA_lib.py:1469-1852): per-database Pearson
correlation, RMSE, RMSE after mapping the predictions onto the subjective scale, and the epsilon-insensitive RMSE*
of ITU-T P.1401 (clause 7.5: Eq. 7-27 perceptual error, Eq. 7-29 degrees-of-freedom correction), per file and per
condition.  Same function names, arguments, returned frame / dict keys and printed lines as the reference, so that
``run_evaluate.py`` style scripts work unchanged; arithmetic is float64 numpy like the reference.

Differences, both deliberate:
  * per-condition means use only numeric columns (pandas >= 2 raises on the reference's ``groupby('con').mean()``
    when the frame carries string columns such as ``db`` or the file path; pandas 1 dropped them silently);
  * a database whose subjective scores contain NaN gets NaN per-file metrics AND is skipped for the per-condition
    mapping (the reference reads a stale / undefined ``y_hat`` there).
"""
import numpy as np
import pandas as pd

_FILE_NAN = ('r_p', 'r_s', 'rmse', 'r_p_map', 'r_s_map', 'rmse_map')
_CON_NAN = _FILE_NAN + ('rmse_star_map',)
_DOF = {None: 0, 'first_order': 1, 'second_order': 3, 'third_order_not_monotonic': 4, 'third_order': 4}


def is_const(x):
    """True when Pearson's r is undefined for ``x`` (NL:1469-1475)."""
    x = np.asarray(x, dtype=float)
    m = np.mean(x)
    return bool(np.linalg.norm(x - m) < 1e-13 * np.abs(m) or np.all(x == x[0]))


def calc_rmse(y_true, y_pred, d=0):
    """RMSE with ``d`` degrees of freedom removed (P.1401 Eq. 7-29); NaN when N - d < 1 (NL:1498-1507)."""
    err2 = np.square(np.asarray(y_true, dtype=float) - np.asarray(y_pred, dtype=float))
    if d == 0:
        return np.sqrt(np.mean(err2))
    n = err2.shape[0]
    return np.sqrt(np.sum(err2) / (n - d)) if n - d >= 1 else np.nan


def calc_rmse_star(mos_sub, mos_obj, ci, d):
    """(RMSE*, perceptual error, error): errors inside the confidence interval do not count (NL:1509-1524)."""
    mos_sub, mos_obj = np.asarray(mos_sub, dtype=float), np.asarray(mos_obj, dtype=float)
    error = mos_sub - mos_obj
    if np.isnan(ci).any():
        return np.nan, np.nan, error
    p_error = np.clip(np.abs(error) - ci, 0, None)                 # P.1401 Eq. 7-27
    n = mos_sub.shape[0]
    rmse_star = np.sqrt(np.sum(p_error ** 2) / (n - d)) if n - d >= 1 else np.nan
    return rmse_star, p_error, error


def calc_mapped(x, b):
    """Polynomial b[0] + b[1] x + b[2] x^2 + ... (NL:1526-1532)."""
    x = np.asarray(x, dtype=float)
    return np.vander(x, len(b), increasing=True) @ np.asarray(b, dtype=float)


def _polyfit(y, y_hat, order):
    a = np.vander(np.asarray(y_hat, dtype=float), order + 1, increasing=True)
    return np.linalg.lstsq(a, np.asarray(y, dtype=float), rcond=None)[0]


def fit_first_order(y_con, y_con_hat):
    return _polyfit(y_con, y_con_hat, 1)


def fit_second_order(y_con, y_con_hat):
    return _polyfit(y_con, y_con_hat, 2)


def fit_third_order(y_con, y_con_hat):
    """Unconstrained cubic; says so when it is not monotonic over the range of the predictions (NL:1544-1555)."""
    b = _polyfit(y_con, y_con_hat, 3)
    stationary = np.roots(np.polyder(np.poly1d(b[::-1])))
    stationary = stationary[np.imag(stationary) == 0]
    if not all(np.logical_or(stationary > max(y_con_hat), stationary < min(y_con_hat))):
        print('Not monotonic!!!')
    return b


def _con_mean(dfile_db, column):
    """Per-condition mean of one column, conditions in sorted order (the order of ``groupby``)."""
    return dfile_db.groupby('con')[column].mean().to_numpy()


def fit_monotonic_third_order(dfile_db, dcon_db=None, pred=None, target_mos=None, target_ci=None, mapping=None):
    """Cubic mapping constrained to a non-negative slope on a 0.1 grid over the prediction range, SLSQP from the
    identity (NL:1557-1645).  ``mapping``: 'error' (squared error) or 'pError' (squared perceptual error)."""
    from scipy.optimize import minimize
    if mapping not in ('error', 'pError'):
        raise NotImplementedError
    y_hat = dfile_db[pred].to_numpy()
    ref = dfile_db if dcon_db is None else dcon_db
    target = ref[target_mos].to_numpy()
    ci = ref[target_ci].to_numpy() if target_ci in ref else 0
    grid = np.arange(min(y_hat) - 0.01, max(y_hat) + 0.01, 0.1)
    con = None if dcon_db is None else dfile_db['con'].to_numpy()

    def objective(p):
        x_map = calc_mapped(y_hat, p)
        if con is not None:
            x_map = pd.Series(x_map).groupby(con).mean().to_numpy()
        err = x_map - target
        if mapping == 'pError':
            err = np.clip(np.abs(err) - ci, 0, None)
        return np.sum(err ** 2)

    res = minimize(objective, x0=np.array([0., 1., 0., 0.]), method='SLSQP',
                   constraints=dict(type='ineq', fun=lambda p: p[1] + 2 * p[2] * grid + 3 * p[3] * grid ** 2))
    return res.x


def calc_mapping(dfile_db, mapping=None, dcon_db=None, target_mos=None, target_ci=None, pred=None):
    """-> (polynomial coefficients, degrees of freedom the mapping consumes) (NL:1647-1690)."""
    if mapping not in _DOF:
        raise NotImplementedError
    if dcon_db is not None:
        y, y_hat = dcon_db[target_mos].to_numpy(), _con_mean(dfile_db, pred)
    else:
        y, y_hat = dfile_db[target_mos].to_numpy(), dfile_db[pred].to_numpy()
    if mapping is None:
        b = np.array([0, 1, 0, 0])
    elif mapping == 'first_order':
        b = fit_first_order(y, y_hat)
    elif mapping == 'second_order':
        b = fit_second_order(y, y_hat)
    elif mapping == 'third_order_not_monotonic':
        b = fit_third_order(y, y_hat)
    else:
        b = fit_monotonic_third_order(dfile_db, dcon_db=dcon_db, pred=pred, target_mos=target_mos,
                                      target_ci=target_ci, mapping='error')
    return b, _DOF[mapping]


def calc_eval_metrics(y, y_hat, y_hat_map=None, d=None, ci=None):
    """Pearson r, RMSE, mapped RMSE, mapped RMSE* (NL:1477-1496)."""
    from scipy.stats import pearsonr
    y, y_hat = np.asarray(y, dtype=float), np.asarray(y_hat, dtype=float)
    r = {'r_p': np.nan, 'rmse': np.nan, 'rmse_map': np.nan, 'rmse_star_map': np.nan}
    if not (is_const(y_hat) or np.isnan(y).any()):
        r['r_p'] = pearsonr(y, y_hat)[0]
    r['rmse'] = calc_rmse(y, y_hat)
    if y_hat_map is not None:
        r['rmse_map'] = calc_rmse(y, y_hat_map, d=d)
        if ci is not None:
            r['rmse_star_map'] = calc_rmse_star(y, y_hat_map, ci, d)[0]
    return r


def _scatter(x, y, b, title, xlabel, ylabel, size):
    import matplotlib.pyplot as plt
    xx = np.arange(0, 6, 0.01)
    plt.figure(figsize=(3.0, 3.0), dpi=300)
    plt.clf()
    plt.plot(x, y, 'o', label='Original data', markersize=size)
    plt.plot([0, 5], [0, 5], 'gray')
    plt.plot(xx, calc_mapped(xx, b), 'r', label='Fitted line')
    plt.axis([1, 5, 1, 5])
    plt.gca().set_aspect('equal', adjustable='box')
    plt.grid(True)
    plt.xticks(np.arange(1, 6))
    plt.yticks(np.arange(1, 6))
    plt.title(title)
    plt.ylabel(ylabel)
    plt.xlabel(xlabel)
    plt.show()


def eval_results(df, dcon=None, target_mos='mos', target_ci='mos_ci', pred='mos_pred', mapping=None,
                 do_print=False, do_plot=False):
    """Per-database and overall metrics of ``df[pred]`` against ``df[target_mos]`` (NL:1687-1852).

    ``df`` needs a ``db`` column; with ``dcon`` (per-condition frame: db, con, target[, target_ci]) and a ``con``
    column in ``df`` the per-condition block is filled too and ``df['y_hat_map']`` receives the predictions mapped
    with the per-condition polynomial.  -> (frame with one row per database, dict of overall results).
    """
    rows = []
    df['y_hat_map'] = np.nan
    has_con = False
    for db_name in df.db.astype('category').cat.categories:
        sel = df.db == db_name
        df_db = df.loc[sel]
        dcon_db = dcon.loc[dcon.db == db_name] if dcon is not None else None
        has_con = dcon_db is not None
        y, y_hat = df_db[target_mos].to_numpy(), df_db[pred].to_numpy()
        labelled = not np.isnan(y).any()

        # per file
        if labelled:
            b, d = calc_mapping(df_db, mapping=mapping, target_mos=target_mos, target_ci=target_ci, pred=pred)
            r = calc_eval_metrics(y, y_hat, y_hat_map=calc_mapped(y_hat, b), d=d)
            r.pop('rmse_star_map')
        else:
            r = dict.fromkeys(_FILE_NAN, np.nan)
        row = {'db': db_name}
        row.update({k + '_file': v for k, v in r.items()})

        # per condition
        r_con = dict.fromkeys(_CON_NAN, np.nan)
        with_con = has_con and 'con' in df_db
        if with_con and labelled:
            y_con = dcon_db[target_mos].to_numpy()
            y_con_hat = _con_mean(df_db, pred)
            if not np.isnan(y_con).any():
                ci_con = dcon_db[target_ci].to_numpy() if target_ci in dcon_db else None
                b_con, d = calc_mapping(df_db, dcon_db=dcon_db, mapping=mapping, target_mos=target_mos,
                                        target_ci=target_ci, pred=pred)
                mapped = calc_mapped(y_hat, b_con)
                df.loc[sel, 'y_hat_map'] = mapped
                y_con_hat_map = pd.Series(mapped).groupby(df_db['con'].to_numpy()).mean().to_numpy()
                r_con = calc_eval_metrics(y_con, y_con_hat, y_hat_map=y_con_hat_map, d=d, ci=ci_con)
                if do_plot:
                    _scatter(y_con_hat, y_con, b_con, db_name + ' per con', 'Pred ' + target_mos.upper(),
                             'Sub ' + target_mos.upper(), 3)
        row.update({k + '_con': v for k, v in r_con.items()})
        rows.append(row)

        if do_plot and labelled:
            _scatter(y_hat, y, b, db_name + ' per file', 'Predicted ' + target_mos.upper(),
                     'Subjective ' + target_mos.upper(), 2)
        if do_print and labelled:
            if with_con:
                print('%-30s r_p_file: %0.2f, rmse_map_file: %0.2f, r_p_con: %0.2f, rmse_map_con: %0.2f, rmse_star_map_con: %0.2f'
                      % (db_name + ':', row['r_p_file'], row['rmse_map_file'], row['r_p_con'], row['rmse_map_con'],
                         row['rmse_star_map_con']))
            else:
                print('%-30s r_p_file: %0.2f, rmse_map_file: %0.2f' % (db_name + ':', row['r_p_file'], row['rmse_map_file']))

    db_results_df = pd.DataFrame(rows)
    overall = {}
    total = calc_eval_metrics(df[target_mos].to_numpy(), df[pred].to_numpy())
    overall['r_p_all'], overall['rmse_all'] = total['r_p'], total['rmse']
    for k in ('r_p', 'rmse', 'rmse_map'):
        overall[k + '_mean_file'] = db_results_df[k + '_file'].mean()
    for k in ('r_p', 'rmse', 'rmse_map', 'rmse_star_map'):
        overall[k + '_mean_con'] = db_results_df[k + '_con'].mean() if has_con else np.nan
    return db_results_df, overall
