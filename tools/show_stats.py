"""Print name / calls / average us / share from a rocprofv3 *_kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:72].ljust(72), r['Calls'].rjust(5), "%10.1f us" % (float(r['AverageNs']) / 1e3), r['Percentage'])
