"""TEST INFRASTRUCTURE ONLY — mints tests/golden/clean_caption_cases.json from the reference's own caption cleaner.

    python oracle/make_golden_caption.py       (build container only: needs /root/reference)

OpenSoraPipeline._clean_caption / _basic_clean / text_preprocessing (pipeline_open_sora.py:298-424) are compiled from the
reference file where it lies (the module does not import here: ftfy / bs4 / torchvision are absent).  Their two third-party calls
are stood in for by the identity — ``ftfy.fix_text`` IS the identity on text that is not mojibake, ``BeautifulSoup(s).text`` on
text without '<' or '&' — and the inputs below stay inside that domain, so what is pinned is the reference's rule sequence.
"""
from __future__ import annotations

import ast
import html
import json
import os
import re
import sys
import urllib.parse as ul
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = "/root/reference/videosys/pipelines/open_sora/pipeline_open_sora.py"

CASES = [
    "A Sunset over the SEA",
    "  a cat,sitting on a mat.Then it jumps  ",
    "a drone shot of a city at night, aesthetic score: 6.5.",
    "this-is-my-cute-cat_running_in_the-park",
    "visit https://example.com/page or www.test.org now",
    "photo by @john_doe #123 #1234567 of a dog",
    "file IMG_20200101.jpg shows a bird... and more!!",
    "the “quoted” ‘text’ with — dashes – and ― bars",
    "free shipping worldwide free shipping, click on here to download free",
    "jc6640 abc123def 6640vc231 j2d1a2a page 12",
    "size 1920x1080 and 3.5x2 and 12×8",
    "ip 192.168.0.1 and number 1234567 stay",
    "\"a fully quoted caption\"",
    "'single quoted'",
    "-leading dash and trailing colon:",
    ".hidden",
    "|0| a beautiful day",
    "camera motion: pan right. motion score: 3.2.",
    "a/b testing \\ backslash * star ~ tilde {curly} [square] (round)",
    "word . word  :  colon",
    "湖边的日落 sunset by the lake",
    "a%20url%2Dencoded+caption",
    "multi\\nline caption",
    "<person> walks by",
    "a photo.PNG image of png images",
]


def reference_cleaners(fix_text=lambda t: t, soup_text=lambda s: s):
    """The reference's own caption functions, compiled from its files: -> (open_sora, latte) namespaces with ``_clean_caption(c)``
    and ``text_preprocessing`` / ``_text_preprocessing``.  ``fix_text`` / ``soup_text`` stand in for ftfy.fix_text and
    BeautifulSoup(s, features="html.parser").text (identity by default, see the module docstring)."""
    src = open(PATH).read()
    ns = {"re": re, "html": html, "ul": ul, "ftfy": SimpleNamespace(fix_text=fix_text),
          "BeautifulSoup": lambda s, features=None: SimpleNamespace(text=soup_text(s))}
    funcs = {}
    for node in ast.parse(src).body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "BAD_PUNCT_REGEX":
            exec(compile(ast.Module([node], []), PATH, "exec"), ns)
        if isinstance(node, ast.ClassDef) and node.name == "OpenSoraPipeline":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in ("_clean_caption", "_basic_clean", "text_preprocessing"):
                    item.decorator_list = []
                    exec(compile(ast.Module([item], []), PATH, "exec"), ns)
                    funcs[item.name] = ns[item.name]
    me = SimpleNamespace()
    me._basic_clean = funcs["_basic_clean"]
    me._clean_caption = lambda c: funcs["_clean_caption"](me, c)
    me.text_preprocessing = lambda c, use=True: funcs["text_preprocessing"](me, c, use)
    # LattePipeline: the diffusers IFPipeline copy (pipeline_latte.py:185-189 bad_punct_regex, :519-650)
    lpath = "/root/reference/videosys/pipelines/latte/pipeline_latte.py"
    lns = dict(ns)
    lat = SimpleNamespace()
    for node in ast.parse(open(lpath).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == "LattePipeline":
            for item in node.body:
                if isinstance(item, ast.Assign) and getattr(item.targets[0], "id", None) == "bad_punct_regex":
                    exec(compile(ast.Module([item], []), lpath, "exec"), lns)
                    lat.bad_punct_regex = lns["bad_punct_regex"]
                if isinstance(item, ast.FunctionDef) and item.name in ("_clean_caption", "_text_preprocessing"):
                    exec(compile(ast.Module([item], []), lpath, "exec"), lns)
    lat._clean_caption = lambda c: lns["_clean_caption"](lat, c)
    lat._text_preprocessing = lambda c, clean_caption=False: lns["_text_preprocessing"](lat, c, clean_caption=clean_caption)
    return me, lat


def main():
    me, lat = reference_cleaners()
    out = []
    for c in CASES:
        assert "<" not in c.replace("<person>", "") and "&" not in c
        out.append({"in": c, "once": me._clean_caption(c), "twice": me.text_preprocessing(c),
                    "plain": me.text_preprocessing(c, False),
                    "latte_twice": lat._text_preprocessing(c, clean_caption=True)[0],
                    "latte_plain": lat._text_preprocessing(c, clean_caption=False)[0]})
    with open(os.path.join(ROOT, "tests", "golden", "clean_caption_cases.json"), "w") as fh:
        json.dump(out, fh, ensure_ascii=False, indent=1)
    for o in out[:6]:
        print(repr(o["in"]), "->", repr(o["twice"]))
    print(len(out), "cases")


if __name__ == "__main__":
    main()
