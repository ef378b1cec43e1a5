"""Average PMC counter values per kernel from a `rocprofv3 --kernel-trace --pmc ... --output-format csv` run.
usage: python3 tools/pmc_by_kernel.py <counter_collection.csv> [substring of the kernel name]"""
import collections
import csv
import sys


def main(path, needle=""):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())), "launches=%d" % len(next(iter(cs.values()))))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
