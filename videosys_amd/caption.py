"""Caption cleaning of the Open-Sora pipeline — what ``text_preprocessing(text)`` does to a prompt before the tokenizer sees it
(/root/reference/videosys/pipelines/open_sora/pipeline_open_sora.py:27-29 BAD_PUNCT_REGEX, :298-302 _basic_clean, :304-415
_clean_caption, :417-424 text_preprocessing: the cleaner is applied TWICE).  It is the DeepFloyd-IF training-time cleaner; its
regular expressions are the specification, so they are the same expressions here — held as an ordered rule table instead of a
statement list — and tests/test_host_cpu.py checks the result against outputs minted from the reference's own function
(tests/golden/clean_caption_cases.json, oracle/make_golden_caption.py).

Two steps of the reference call third-party packages that are not in this image:
  * ``BeautifulSoup(caption, features="html.parser").text`` — the text content of the caption parsed as HTML.  bs4's html.parser
    backend IS the standard library's ``html.parser``; ``_html_text`` collects the same character data with it (tags dropped,
    character references resolved).  bs4 is used when importable.
  * ``ftfy.fix_text`` (mojibake repair, width / ligature folding, NFC).  Used when importable; otherwise NFC normalisation only —
    identical on text that is not broken (every ASCII prompt), documented as the one difference on text that is.
"""
from __future__ import annotations

import html
import re
import unicodedata
import urllib.parse as ul
from html.parser import HTMLParser

_URL_TLDS = r"(?:com|co|ru|net|org|edu|gov|it)"
_IMG_EXT = r"(?:png|jpg|jpeg|bmp|webp|eps|pdf|apk|mp4)"

BAD_PUNCT = re.compile("[" + re.escape("#®•©™&@·º½¾¿¡§~)(][}{|\\/*") + "]{1,}")   # :27-29

# (pattern, replacement) applied in this order BEFORE the html step (:308-318)
_PRE_HTML = [
    (r"<person>", "person"),
    (r"\b((?:https?:(?:\/{1,3}|[a-zA-Z0-9%])|[a-zA-Z0-9.\-]+[.]" + _URL_TLDS + r"[\w/-]*\b\/?(?!@)))", ""),
    (r"\b((?:www:(?:\/{1,3}|[a-zA-Z0-9%])|[a-zA-Z0-9.\-]+[.]" + _URL_TLDS + r"[\w/-]*\b\/?(?!@)))", ""),
]

# after the html step, up to the dash / underscore rule (:323-381)
_MID = [
    (r"@[\w\d]+\b", ""),                       # @nickname
    (r"[\u31c0-\u31ef]+", ""),                 # CJK strokes
    (r"[\u31f0-\u31ff]+", ""),                 # katakana phonetic extensions
    (r"[\u3200-\u32ff]+", ""),                 # enclosed CJK letters and months
    (r"[\u3300-\u33ff]+", ""),                 # CJK compatibility
    (r"[\u3400-\u4dbf]+", ""),                 # CJK unified ideographs extension A
    (r"[\u4dc0-\u4dff]+", ""),                 # Yijing hexagram symbols
    (r"[\u4e00-\u9fff]+", ""),                 # CJK unified ideographs
    (r"[\u002D\u058A\u05BE\u1400\u1806\u2010-\u2015\u2E17\u2E1A\u2E3A\u2E3B\u2E40\u301C\u3030\u30A0\uFE31\uFE32\uFE58\uFE63\uFF0D]+", "-"),
    (r"[`´«»“”¨]", '"'),
    (r"[‘’]", "'"),
    (r"&quot;?", ""),
    (r"&amp", ""),
    (r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}", " "),   # ip addresses
    (r"\d:\d\d\s+$", ""),                       # article ids
    (r"\\n", " "),
    (r"#\d{1,3}\b", ""),
    (r"#\d{5,}\b", ""),
    (r"\b\d{6,}\b", ""),
    (r"[\S]+\." + _IMG_EXT, ""),                # file names
    (r"[\"\']{2,}", '"'),
    (r"[\.]{2,}", " "),
    (BAD_PUNCT, " "),
    (r"\s+\.\s+", " "),
]

# after _basic_clean (:390-413)
_POST = [
    (r"\b[a-zA-Z]{1,3}\d{3,15}\b", ""),
    (r"\b[a-zA-Z]+\d+[a-zA-Z]+\b", ""),
    (r"\b\d+[a-zA-Z]+\d+\b", ""),
    (r"(worldwide\s+)?(free\s+)?shipping", ""),
    (r"(free\s)?download(\sfree)?", ""),
    (r"\bclick\b\s(?:for|on)\s\w+", ""),
    (r"\b" + _IMG_EXT + r"(\simage[s]?)?", ""),
    (r"\bpage\s+\d+\b", ""),
    (r"\b\d*[a-zA-Z]+\d+[a-zA-Z]+\d+[a-zA-Z\d]*\b", " "),
    (r"\b\d+\.?\d*[xх×]\d+\.?\d*\b", ""),
    (r"\b\s+\:\s+", ": "),
    (r"(\D[,\./])\b", r"\1 "),
    (r"\s+", " "),
    (r"^[\"\']([\w\W]+)[\"\']$", r"\1"),
    (r"^[\'\_,\-\:;]", ""),
    (r"[\'\_,\-\:\-\+]$", ""),
    (r"^\.\S+$", ""),
]

_compile = lambda rules: [(p if isinstance(p, re.Pattern) else re.compile(p), r) for p, r in rules]
_PRE_HTML, _MID, _POST = _compile(_PRE_HTML), _compile(_MID), _compile(_POST)
_DASHES = re.compile(r"(?:\-|\_)")


class _Text(HTMLParser):
    def __init__(self):
        super().__init__(convert_charrefs=True)
        self.parts = []

    def handle_data(self, data):
        self.parts.append(data)


def _html_text(s: str) -> str:
    try:
        from bs4 import BeautifulSoup

        return BeautifulSoup(s, features="html.parser").text
    except ImportError:
        p = _Text()
        p.feed(s)
        p.close()
        return "".join(p.parts)


def basic_clean(text: str, strip: bool = True) -> str:
    """_basic_clean (:298-302).  ``strip=False``: the inline form of LattePipeline._clean_caption (pipeline_latte.py:617-618), which
    does not strip at this point."""
    try:
        import ftfy

        text = ftfy.fix_text(text)
    except ImportError:
        text = unicodedata.normalize("NFC", text)
    text = html.unescape(html.unescape(text))
    return text.strip() if strip else text


def clean_caption(caption, mid_strip: bool = True) -> str:
    """_clean_caption (:304-415), one application.  ``mid_strip=False`` = LattePipeline._clean_caption (pipeline_latte.py:534-650:
    the same rule sequence, copied from diffusers' IFPipeline, without the strip inside the ftfy / unescape step)."""
    s = ul.unquote_plus(str(caption)).strip().lower()
    for pat, rep in _PRE_HTML:
        s = pat.sub(rep, s)
    s = _html_text(s)
    for pat, rep in _MID:
        s = pat.sub(rep, s)
    if len(_DASHES.findall(s)) > 3:           # this-is-my-cute-cat / this_is_my_cute_cat
        s = _DASHES.sub(" ", s)
    s = basic_clean(s, mid_strip)
    for pat, rep in _POST:
        s = pat.sub(rep, s)
    return s.strip()


def text_preprocessing(text: str, use_text_preprocessing: bool = True, mid_strip: bool = True) -> str:
    """text_preprocessing (:417-424; LattePipeline._text_preprocessing, pipeline_latte.py:519-531): the cleaner twice (as at training
    time), or lower-case + strip."""
    if use_text_preprocessing:
        return clean_caption(clean_caption(text, mid_strip), mid_strip)
    return text.lower().strip()
