"""Group a `rocprofv3 --kernel-trace --output-format csv` trace by (kernel, grid size): which launch SHAPES of one kernel
template take the time (the --stats summary merges them).  usage: python3 tools/trace_by_shape.py <kernel_trace.csv> [top]"""
import collections
import csv
import sys


def main(path, top=40):
    acc = collections.defaultdict(lambda: [0, 0])
    total = 0
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:70]
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
        acc[key][0] += dur
        acc[key][1] += 1
        total += dur
    print("total kernel time %.1f ms" % (total / 1e6))
    for (name, wgs), (ns, calls) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:top]:
        print("%6.2f%% %9.2f ms %6d calls %9.1f us/call  wgs=%-7d %s" % (100.0 * ns / total, ns / 1e6, calls, ns / 1e3 / calls, wgs, name))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
