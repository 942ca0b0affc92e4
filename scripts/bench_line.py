import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    j = json.loads(line)
    print("streams", j["config"]["streams_per_gpu"], "value %.0f hyp/s" % j["value"], "ms/step %.4f" % j["ms_per_step"], "k2 us %.1f" % j["roofline"]["avg_launch_us"], "%.0f GB/s" % j["roofline"]["achieved"], "frac %.3f" % j["roofline"]["frac"])
