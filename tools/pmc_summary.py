"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel."""
import glob, sys
import pandas as pd
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        t = pd.read_csv(f)
        t = t[~t.Kernel_Name.str.contains('at::|rocclr')]
        t['k'] = t.Kernel_Name.str.split('(').str[0]
        print(t.groupby(['k', 'Counter_Name'])['Counter_Value'].mean().unstack(0).to_string())
