"""Caption cleaning (video-llamagen_amd/caption.py) against hand-derived vectors.

The reference's T5Embedder.clean_caption / text_preprocessing (language/t5.py:83-200) cannot be imported in this container (ftfy and bs4 are
absent), so every expected string below was derived by applying the reference's rules BY HAND, in order, to the input (the comments name the
rules that fire).  Two passes unless stated."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("vlg_caption", os.path.join(ROOT, "video-llamagen_amd", "caption.py"))
cap = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cap)

CASES = [
    # plain text: lower + strip only
    ("  A Cat Sitting on a Mat  ", "a cat sitting on a mat"),
    # url rule: "https://" and "example.com/page" are two matches of the same alternation; the blanks collapse later
    ("Visit https://example.com/page now", "visit now"),
    ("see www.shop.co/item-12 today", "see today"),
    # @nickname, "#12", file name
    ("photo by @john_doe #12 IMG_1234.jpg", "photo by"),
    # more than three dashes / underscores -> blanks; three or fewer stay
    ("this-is-my-cute-cat", "this is my cute cat"),
    ("a_b_c_d_e", "a b c d e"),
    ("a-b-c", "a-b-c"),
    # every kind of dash becomes "-"
    ("long—dash and–dash", "long-dash and-dash"),
    # html: tags dropped, references decoded by the parser (so the later '&quot;' / '&amp' rules see nothing); '&' is bad punctuation
    ("<b>Bold</b> &amp; <i>italic</i> text", "bold italic text"),
    ('&quot;Sale&quot; 50% off!!! ***AUSVERKAUFT***', '"sale" 50% off!!! ausverkauft'),
    # ip address -> blank
    ("Price: 192.168.1.1 server", "price: server"),
    # typographic quotes -> ascii, then ONE pair of enclosing quotes is peeled off
    ("“Hello” ‘world’", "hello\" 'world"),
    ('"a framed picture"', "a framed picture"),
    # percent-encoding and '+' (urllib.parse.unquote_plus)
    ("cute%20cat+photo", "cute cat photo"),
    # CJK unified ideographs are removed
    ("猫 cat 日本語", "cat"),
    # runs of dots -> blank
    ("Hello... World.. ok", "hello world ok"),
    # '/' is bad punctuation (-> blank); then a blank is inserted after ',' and '.' in front of a word
    ("value,next.word/and", "value, next. word and"),
    # catalogue codes
    ("jc6640 model abc123def", "model"),
    ("6640vc231 unit j2d1a2a", "unit"),
    # sizes (latin x) - matched as digits-letters-digits already - and "page N"
    ("Size 1024x768 image.png page 3", "size"),
    ("poster 30×40 cm", "poster cm"),
    # shop noise
    ("Free Shipping worldwide free shipping Download free", ""),
    ("click for details now", "now"),
    # long digit runs, '#12345', '12:30 ' article id at the end is only removed with trailing blanks inside the string
    ("order 1234567 ref #123456 ok", "order ref ok"),
    # leading / trailing leftovers
    ("-leading dash and trailing+", "leading dash and trailing"),
    # literal backslash-n
    ("line one\\nline two", "line one line two"),
    # <person> placeholder
    ("a <person> walking", "a person walking"),
]


@pytest.mark.parametrize("raw,want", CASES)
def test_two_pass_cleaning(raw, want):
    assert cap.text_preprocessing(raw) == want


def test_without_preprocessing_is_lower_strip():
    assert cap.text_preprocessing("  MiXed Case ", use_text_preprocessing=False) == "mixed case"


def test_cleaning_is_idempotent_after_two_passes():
    for raw, _ in CASES:
        once = cap.text_preprocessing(raw)
        assert cap.text_preprocessing(once) == once, raw


def test_strip_html_matches_parser_text_semantics():
    assert cap.strip_html("a<br>b &lt;c&gt; <span class='x'>d</span>") == "ab <c> d"
    assert cap.strip_html("no tags") == "no tags"


def test_bad_punct_class():
    # the characters the reference's bad_punct_regex lists, among them the backslash and both kinds of brackets
    for ch in "#®•©™&@·º½¾¿¡§~)(][}{|\\/*":
        assert cap.BAD_PUNCT.sub(" ", "a" + ch + "b") == "a b", ch
    assert cap.BAD_PUNCT.sub(" ", "a+b-c") == "a+b-c"
