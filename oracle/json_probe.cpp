// oracle/json_probe.cpp -- TEST INFRASTRUCTURE.  Our driver around the reference's vendored nlohmann/json.hpp
// (compiled from /root/reference/include where it lies, output only under oracle/_ref/).  It builds the two JSON
// documents with exactly the statements the reference uses, so the bytes nlohmann emits become a golden for the
// product's own emitter:
//   size   <raw_filename> <w> <h> <outW> <outH>      -> src/preprocess.cpp:126-134  (compact, one line)
//   poly   <base_name> <orig_w> <orig_h>  (stdin: n, then per contour: m, then m pairs "x y")
//                                                    -> src/mask2polygon.cpp:68-109 (setw(4))
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "nlohmann/json.hpp"

using json = nlohmann::json;

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::string mode = argv[1];
    if (mode == "size" && argc == 7) {
        json j;
        j[argv[2]] = { { "original_width", std::stoi(argv[3]) },
                       { "original_height", std::stoi(argv[4]) },
                       { "scaled_width", std::stoi(argv[5]) },
                       { "scaled_height", std::stoi(argv[6]) } };
        std::cout << j << std::endl;
        return 0;
    }
    if (mode == "poly" && argc == 5) {
        std::string base_name = argv[2];
        json j;
        j["version"] = "1.0.2.812";
        j["imagePath"] = base_name + ".raw";
        j["imageData"] = nullptr;
        j["flags"] = json::object();
        j["shapes"] = json::array();
        int n = 0;
        std::cin >> n;
        for (int c = 0; c < n; ++c) {
            int m = 0;
            std::cin >> m;
            json shape;
            shape["label"] = 1;
            shape["labelIndex"] = 0;
            json points;
            for (int k = 0; k < m; ++k) {
                int x, y;
                std::cin >> x >> y;
                points.push_back({ x, y });
            }
            shape["points"] = points;
            shape["shape_type"] = "polygon";
            shape["description"] = "";
            shape["mask"] = nullptr;
            shape["group_id"] = nullptr;
            shape["flags"] = json::object();
            j["shapes"].push_back(shape);
        }
        j["imageWidth"] = std::stoi(argv[3]);
        j["imageHeight"] = std::stoi(argv[4]);
        std::cout << std::setw(4) << j << std::endl;
        return 0;
    }
    return 2;
}
