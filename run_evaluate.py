# -*- coding: utf-8 -*-
"""Drop-in for the reference's ``run_evaluate.py`` (reference run_evaluate.py:13-38): predict a labelled corpus from a
CSV with the HIP path, then print the per-database / overall P.1401 statistics.  The reference hard-codes its
argument dict in the script; here the same keys come from the command line (defaults = the reference's values).

If a ``--csv_con`` per-condition CSV is given, both CSVs need a ``con`` column; without it only per-file results
are calculated.  Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N`` (clips are sharded,
every rank holds the gathered predictions, rank 0 prints).
"""
import argparse

from nisqa_amd.NISQA_model import nisqaModel


def build_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--pretrained_model', default='weights/nisqa.tar', type=str)
    p.add_argument('--data_dir', required=True, type=str, help='corpus root (CSV paths are relative to it)')
    p.add_argument('--output_dir', default=None, type=str, help='folder for NISQA_results.csv')
    p.add_argument('--csv_file', default='NISQA_corpus_file.csv', type=str)
    p.add_argument('--csv_con', default=None, type=str, help='per-condition CSV, e.g. NISQA_corpus_con.csv')
    p.add_argument('--csv_deg', default='filepath_deg', type=str)
    p.add_argument('--csv_mos_val', default='mos', type=str)
    p.add_argument('--num_workers', default=6, type=int)
    p.add_argument('--bs', default=40, type=int)
    p.add_argument('--ms_channel', default=None, type=int)
    p.add_argument('--mapping', default='first_order', type=str,
                   choices=['none', 'first_order', 'second_order', 'third_order', 'third_order_not_monotonic'])
    p.add_argument('--do_plot', action='store_true')
    a = vars(p.parse_args(argv))
    args = {'mode': 'predict_csv', 'pretrained_model': a['pretrained_model'], 'data_dir': a['data_dir'],
            'output_dir': a['output_dir'], 'csv_file': a['csv_file'], 'csv_deg': a['csv_deg'],
            'csv_mos_val': a['csv_mos_val'], 'tr_num_workers': a['num_workers'], 'tr_bs_val': a['bs'],
            'ms_channel': a['ms_channel']}
    if a['csv_con']:
        args['csv_con'] = a['csv_con']                 # the reference tests "'csv_con' in args" (NISQA_model.py:817)
    return args, (None if a['mapping'] == 'none' else a['mapping']), a['do_plot']


if __name__ == "__main__":
    import os
    import torch
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not torch.distributed.is_initialized():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        torch.distributed.init_process_group('nccl')
    args, mapping, do_plot = build_args()
    nisqa = nisqaModel(args)
    nisqa.predict()
    if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
        nisqa.evaluate(mapping=mapping, do_print=True, do_plot=do_plot)
