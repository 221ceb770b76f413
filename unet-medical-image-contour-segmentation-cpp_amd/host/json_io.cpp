#include "json_io.h"

#include <cstdio>
#include <stdexcept>

namespace medseg {

std::string json_escape(const std::string &s)
{
    std::string o;
    o.reserve(s.size() + 2);
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20) {                           // nlohmann escapes control characters as \u00XX (DEL passes when ensure_ascii is off)
                char b[8];
                snprintf(b, sizeof b, "\\u%04x", c);
                o += b;
            } else {
                o += (char)c;                         // UTF-8 passes through (ensure_ascii = false)
            }
        }
    }
    return o;
}

std::string size_json_text(const std::string &raw_filename, int w, int h, int scaled_w, int scaled_h)
{
    std::string o = "{\"" + json_escape(raw_filename) + "\":{";
    o += "\"original_height\":" + std::to_string(h);
    o += ",\"original_width\":" + std::to_string(w);
    o += ",\"scaled_height\":" + std::to_string(scaled_h);
    o += ",\"scaled_width\":" + std::to_string(scaled_w);
    o += "}}\n";
    return o;
}

std::string polygon_json_text(const std::vector<Contour> &contours, const std::string &base_name, int original_width,
                              int original_height)
{
    std::string o;
    o += "{\n";
    o += "    \"flags\": {},\n";
    o += "    \"imageData\": null,\n";
    o += "    \"imageHeight\": " + std::to_string(original_height) + ",\n";
    o += "    \"imagePath\": \"" + json_escape(base_name + ".raw") + "\",\n";       // always .raw (src/mask2polygon.cpp:76)
    o += "    \"imageWidth\": " + std::to_string(original_width) + ",\n";
    if (contours.empty()) {
        o += "    \"shapes\": [],\n";
    } else {
        o += "    \"shapes\": [\n";
        for (size_t c = 0; c < contours.size(); ++c) {
            o += "        {\n";
            o += "            \"description\": \"\",\n";
            o += "            \"flags\": {},\n";
            o += "            \"group_id\": null,\n";
            o += "            \"label\": 1,\n";
            o += "            \"labelIndex\": 0,\n";
            o += "            \"mask\": null,\n";
            if (contours[c].empty()) {
                o += "            \"points\": null,\n";      // a default-constructed nlohmann::json that was never push_back'ed
            } else {
                o += "            \"points\": [\n";
                for (size_t k = 0; k < contours[c].size(); ++k) {
                    o += "                [\n";
                    o += "                    " + std::to_string(contours[c][k].x) + ",\n";
                    o += "                    " + std::to_string(contours[c][k].y) + "\n";
                    o += (k + 1 < contours[c].size()) ? "                ],\n" : "                ]\n";
                }
                o += "            ],\n";
            }
            o += "            \"shape_type\": \"polygon\"\n";
            o += (c + 1 < contours.size()) ? "        },\n" : "        }\n";
        }
        o += "    ],\n";
    }
    o += "    \"version\": \"1.0.2.812\"\n";
    o += "}\n";
    return o;
}

namespace {
struct Parser {
    const std::string &s;
    size_t i = 0;
    explicit Parser(const std::string &t) : s(t) {}
    [[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("size JSON parse error: ") + what); }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\r' || s[i] == '\t')) ++i; }
    void expect(char c) { ws(); if (i >= s.size() || s[i] != c) bad("unexpected character"); ++i; }
    bool peek(char c) { ws(); return i < s.size() && s[i] == c; }
    std::string str()
    {
        expect('"');
        std::string o;
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\') {
                if (++i >= s.size()) bad("dangling escape");
                switch (s[i]) {
                case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
                case 'b': o += '\b'; break; case 'f': o += '\f'; break;
                case 'u': {
                    if (i + 4 >= s.size()) bad("short \\u escape");
                    const unsigned cp = (unsigned)std::stoul(s.substr(i + 1, 4), nullptr, 16);
                    if (cp < 0x80) o += (char)cp;
                    else if (cp < 0x800) { o += (char)(0xC0 | cp >> 6); o += (char)(0x80 | (cp & 0x3F)); }
                    else { o += (char)(0xE0 | cp >> 12); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
                    i += 4;
                    break;
                }
                default: o += s[i];
                }
                ++i;
            } else {
                o += s[i++];
            }
        }
        expect('"');
        return o;
    }
    long long integer()
    {
        ws();
        size_t j = i;
        if (j < s.size() && s[j] == '-') ++j;
        while (j < s.size() && s[j] >= '0' && s[j] <= '9') ++j;
        if (j == i) bad("expected an integer");
        const long long v = std::stoll(s.substr(i, j - i));
        i = j;
        return v;
    }
};
}  // namespace

std::map<std::string, std::map<std::string, long long>> parse_size_json(const std::string &text)
{
    Parser p(text);
    std::map<std::string, std::map<std::string, long long>> out;
    p.expect('{');
    if (!p.peek('}')) {
        for (;;) {
            const std::string key = p.str();
            p.expect(':');
            p.expect('{');
            auto &inner = out[key];
            if (!p.peek('}')) {
                for (;;) {
                    const std::string k = p.str();
                    p.expect(':');
                    inner[k] = p.integer();
                    if (p.peek(',')) { p.expect(','); continue; }
                    break;
                }
            }
            p.expect('}');
            if (p.peek(',')) { p.expect(','); continue; }
            break;
        }
    }
    p.expect('}');
    return out;
}

}  // namespace medseg
