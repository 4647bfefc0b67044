// swift-tools-version:5.7
// Headless SwiftPM package that puts librtmi.so behind the reference's raytraceWorld call.
// NOT compiled or tested in this repository's container (no Swift toolchain there); it is what
// a maintainer of nealsid/RayTracer would add.  Build where `swift` exists:
//   make -C ../raytracer_amd/csrc && swift build -Xlinker -L../raytracer_amd -Xlinker -rpath -Xlinker $PWD/../raytracer_amd
import PackageDescription

let package = Package(
    name: "RtmiSwift",
    products: [.library(name: "RtmiSwift", targets: ["RtmiSwift"])],
    targets: [
        .systemLibrary(name: "CRtmi", path: "Sources/CRtmi"),
        .target(name: "RtmiSwift", dependencies: ["CRtmi"]),
    ]
)
