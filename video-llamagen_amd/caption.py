"""Caption cleaning in front of the T5 tokenizer: the behaviour of T5Embedder.text_preprocessing / clean_caption / basic_clean in the
reference (language/t5.py:83-200, the DeepFloyd-IF / PixArt training-time cleaning; two passes when use_text_preprocessing is on, otherwise
lower().strip()).

Host-side string work: no kernel, no torch.  The reference needs two third-party packages for two of its steps; this module does not:
  * `BeautifulSoup(caption, 'html.parser').text` -> the standard library's html.parser (the same tokenizer bs4 drives in that mode):
    tag text is dropped, character references are decoded, <script>/<style> bodies are kept as text exactly as bs4's .text keeps them;
  * `ftfy.fix_text` -> used when the package is importable; without it the step only applies Unicode NFC.  Mojibake repair is the one
    thing then missing; every caption made of well-formed text cleans identically.
The order of the steps is the reference's; each rule below names what it removes.  tests/test_caption_cpu.py pins the module with
hand-derived vectors (the reference module itself cannot be imported here: ftfy and bs4 are absent)."""
import html
import re
import unicodedata
import urllib.parse
from html.parser import HTMLParser

try:  # optional, as in the reference's requirements
    import ftfy as _ftfy
except Exception:  # noqa: BLE001
    _ftfy = None

_IMG_EXT = r"(?:png|jpg|jpeg|bmp|webp|eps|pdf|apk|mp4)"
_TLD = r"(?:com|co|ru|net|org|edu|gov|it)"


def _url_rule(scheme):
    # scheme://... or bare host.tld/path, not followed by '@' (language/t5.py:92-97)
    return re.compile(r"\b((?:" + scheme + r":(?:\/{1,3}|[a-zA-Z0-9%])|[a-zA-Z0-9.\-]+[.]" + _TLD + r"[\w/-]*\b\/?(?!@)))")


class _TextOnly(HTMLParser):
    def __init__(self):
        super().__init__(convert_charrefs=True)
        self.parts = []

    def handle_data(self, data):
        self.parts.append(data)


def strip_html(text):
    """What BeautifulSoup(text, features='html.parser').text returns: the character data between tags, references decoded."""
    p = _TextOnly()
    p.feed(text)
    p.close()
    return "".join(p.parts)


# (pattern, replacement) in the reference's order.  Stage A runs before the dash-count test, stage B after basic_clean.
_CJK_BLOCKS = ((0x31C0, 0x31EF), (0x31F0, 0x31FF), (0x3200, 0x32FF), (0x3300, 0x33FF), (0x3400, 0x4DBF), (0x4DC0, 0x4DFF), (0x4E00, 0x9FFF))
_DASHES = "\u002D\u058A\u05BE\u1400\u1806\u2010-\u2015\u2E17\u2E1A\u2E3A\u2E3B\u2E40\u301C\u3030\u30A0\uFE31\uFE32\uFE58\uFE63\uFF0D"
BAD_PUNCT = re.compile(r"[" + "#®•©™&@·º½¾¿¡§~" + r"\)" + r"\(" + r"\]" + r"\[" + r"\}" + r"\{" + r"\|" + "\\\\" + r"\/" + r"\*" + r"]{1,}")

_STAGE_A = [
    (re.compile(r"@[\w\d]+\b"), ""),                                         # @nickname
    *[(re.compile("[%s-%s]+" % (chr(lo), chr(hi))), "") for lo, hi in _CJK_BLOCKS],   # CJK strokes ... unified ideographs
    (re.compile("[" + _DASHES + "]+"), "-"),                                 # every kind of dash -> "-"
    (re.compile(r"[`´«»“”¨]"), '"'),                                          # quotes to one standard
    (re.compile(r"[‘’]"), "'"),
    (re.compile(r"&quot;?"), ""),
    (re.compile(r"&amp"), ""),
    (re.compile(r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}"), " "),                # ip addresses
    (re.compile(r"\d:\d\d\s+$"), ""),                                        # article ids
    (re.compile(r"\\n"), " "),                                               # a literal backslash-n
    (re.compile(r"#\d{1,3}\b"), ""),                                         # "#123"
    (re.compile(r"#\d{5,}\b"), ""),                                          # "#12345.."
    (re.compile(r"\b\d{6,}\b"), ""),                                         # "123456.."
    (re.compile(r"[\S]+\." + _IMG_EXT), ""),                                 # file names
    (re.compile(r"[\"\']{2,}"), '"'),                                        # runs of quotes
    (re.compile(r"[\.]{2,}"), " "),                                          # runs of dots
    (BAD_PUNCT, " "),                                                        # ***X***, #X ...
    (re.compile(r"\s+\.\s+"), " "),                                          # " . "
]
_STAGE_B = [
    (re.compile(r"\b[a-zA-Z]{1,3}\d{3,15}\b"), ""),                          # jc6640
    (re.compile(r"\b[a-zA-Z]+\d+[a-zA-Z]+\b"), ""),                          # jc6640vc
    (re.compile(r"\b\d+[a-zA-Z]+\d+\b"), ""),                                # 6640vc231
    (re.compile(r"(worldwide\s+)?(free\s+)?shipping"), ""),
    (re.compile(r"(free\s)?download(\sfree)?"), ""),
    (re.compile(r"\bclick\b\s(?:for|on)\s\w+"), ""),
    (re.compile(r"\b" + _IMG_EXT + r"(\simage[s]?)?"), ""),
    (re.compile(r"\bpage\s+\d+\b"), ""),
    (re.compile(r"\b\d*[a-zA-Z]+\d+[a-zA-Z]+\d+[a-zA-Z\d]*\b"), " "),        # j2d1a2a...
    (re.compile(r"\b\d+\.?\d*[xх×]\d+\.?\d*\b"), ""),                         # 1024x768 (latin x, cyrillic х, ×)
    (re.compile(r"\b\s+\:\s+"), ": "),
    (re.compile(r"(\D[,\./])\b"), r"\1 "),                                   # a space after , . / in front of a word
    (re.compile(r"\s+"), " "),
]
_STAGE_C = [
    (re.compile(r"^[\"\']([\w\W]+)[\"\']$"), r"\1"),                         # one pair of enclosing quotes
    (re.compile(r"^[\'\_,\-\:;]"), ""),                                      # leading / trailing leftovers
    (re.compile(r"[\'\_,\-\:\-\+]$"), ""),
    (re.compile(r"^\.\S+$"), ""),                                            # ".word" alone
]
_SEPARATORS = re.compile(r"(?:\-|\_)")


def basic_clean(text):
    """language/t5.py:92-95: ftfy.fix_text, two rounds of html.unescape, strip."""
    text = _ftfy.fix_text(text) if _ftfy is not None else unicodedata.normalize("NFC", text)
    return html.unescape(html.unescape(text)).strip()


def clean_caption(caption):
    """One pass of the reference's clean_caption (language/t5.py:97-200)."""
    c = urllib.parse.unquote_plus(str(caption)).strip().lower()
    c = c.replace("<person>", "person")
    c = _url_rule("https?").sub("", c)
    c = _url_rule("www").sub("", c)
    c = strip_html(c)
    for pat, rep in _STAGE_A:
        c = pat.sub(rep, c)
    if len(_SEPARATORS.findall(c)) > 3:          # this-is-my-cute-cat / this_is_my_cute_cat
        c = _SEPARATORS.sub(" ", c)
    c = basic_clean(c)
    for pat, rep in _STAGE_B:
        c = pat.sub(rep, c)
    # (the reference calls caption.strip() here and drops the result: the string keeps its outer blanks for the next four rules)
    for pat, rep in _STAGE_C:
        c = pat.sub(rep, c)
    return c.strip()


def text_preprocessing(text, use_text_preprocessing=True):
    """language/t5.py:83-90: the cleaning is applied TWICE ("the exact text cleaning as was in the training stage")."""
    if use_text_preprocessing:
        return clean_caption(clean_caption(text))
    return text.lower().strip()
