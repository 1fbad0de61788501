#!/usr/bin/env python3
"""Extracts small constant tables of the AC context model from the reference's SOURCE (numbers the format fixes) into
tests/golden/ref_constant_tables.json, so that a test can catch a transcription error in this repository's copies -
encoder, oracle and product all carry their own, and a shared typo would pass every self-consistency test:
  kCoeffFreqContext, kCoeffNumNonzeroContext, BlockCtxMap::kDefaultCtxMap   lib/jxl/ac_context.h:29-42,91-96
  kStrategyOrder                                                            lib/jxl/coeff_order.h:44-46
  covered_blocks_x / _y (their kLut tables)                                 lib/jxl/ac_strategy.h:148-164
  kAcStrategyToQuantTableMap (as indices into enum QuantTable)              lib/jxl/quant_weights.h:304-348
usage (where /root/reference exists): python tests/golden/make_tables_golden.py"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/lib/jxl"


def array(path, name):
    text = open(os.path.join(REF, path)).read()
    m = re.search(r"%s(?:\[[^\]]*\])?\s*=\s*\{(.*?)\};" % re.escape(name), text, re.S)
    body = re.sub(r"//.*", "", m.group(1))
    return [int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\d+", body)]


def main():
    out = {"kCoeffFreqContext": array("ac_context.h", "kCoeffFreqContext"),
           "kCoeffNumNonzeroContext": array("ac_context.h", "kCoeffNumNonzeroContext"),
           "kDefaultCtxMap": array("ac_context.h", "kDefaultCtxMap"),
           "kStrategyOrder": array("coeff_order.h", "kStrategyOrder")}
    text = open(os.path.join(REF, "ac_strategy.h")).read()
    luts = re.findall(r"kLut\[\]\s*=\s*\{(.*?)\};", text, re.S)
    out["covered_blocks_x"] = [int(t) for t in re.findall(r"\d+", luts[0])]
    out["covered_blocks_y"] = [int(t) for t in re.findall(r"\d+", luts[1])]
    q = open(os.path.join(REF, "quant_weights.h")).read()
    enum_body = re.sub(r"//.*", "", re.search(r"enum class QuantTable[^{]*\{(.*?)\}", q, re.S).group(1))
    names = [n.split("=")[0].strip() for n in enum_body.split(",") if n.strip()]
    mapping = re.search(r"kAcStrategyToQuantTableMap\s*=\s*\{(.*?)\};", q, re.S).group(1)
    out["strategy_to_quant_table"] = [names.index(n) for n in re.findall(r"QuantTable::(\w+)", mapping)]
    assert [len(v) for v in out.values()] == [64, 64, 39, 27, 27, 27, 27], [len(v) for v in out.values()]
    json.dump(out, open(os.path.join(HERE, "ref_constant_tables.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
