#!/usr/bin/env python3
"""Extracts the reference's own ICC codec test vector (lib/jxl/icc_codec_test.cc:52-211: kTestProfile and its coded form
kEncodedTestProfile, which the reference test decodes and compares byte for byte) into two binary fixtures:
  ref_icc_test_profile.icc   the profile
  ref_icc_test_profile.enc   the coded stream as test::ReadICC reads it (U64 size, histograms, ANS data; starts at bit 0)
usage (where /root/reference exists): python tests/golden/make_icc_golden.py"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/lib/jxl/icc_codec_test.cc"


def array(text, name):
    m = re.search(r"%s\[\]\s*=\s*\{(.*?)\};" % re.escape(name), text, re.S)
    body = re.sub(r"//.*", "", m.group(1))
    return bytes(int(t, 0) for t in re.findall(r"0x[0-9a-fA-F]+|\d+", body))


def main():
    text = open(SRC).read()
    icc, enc = array(text, "kTestProfile"), array(text, "kEncodedTestProfile")
    open(os.path.join(HERE, "ref_icc_test_profile.icc"), "wb").write(icc)
    open(os.path.join(HERE, "ref_icc_test_profile.enc"), "wb").write(enc)
    print(len(icc), len(enc))


if __name__ == "__main__":
    sys.exit(main())
