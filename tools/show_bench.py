import json, sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["bound"], r["frac"], "traffic", r["traffic"], r["traffic_source"], r["traffic_stale"], r["traffic_vs_compulsory"], r["valu_issue_busy"])
print("fit traffic", (d["fit"]["roofline"]["traffic"] or {}).get("source"))
print("structured", d["structured"]["ms_per_step"], d["structured"]["roofline"]["traffic"], d["structured"]["roofline"]["traffic_vs_compulsory"], d["structured"]["roofline"]["traffic_stale"])
print("c4", d["c4"]["value"], d["c4"]["ms_per_step"])
for x in d["streaming"]["open_loop"]["runs"]: print(x["fit_mode"], x["arrival_rate_per_sec"], round(x["sustained_interactions_per_sec"]), round(x["update_to_visible_ms"]["p50"]), x["keeps_up"])
