// Swift face of librtmi.so for the reference's types.  NOT compiled in this repository's
// container (no Swift toolchain); written against include/rtmi.h.
//
// The value types below carry exactly the stored properties of the reference's classes
// (Renderables.swift, ObjLoader.swift, Utilities.swift); a maintainer would pass the reference's own
// objects instead and delete these.
import CRtmi

public struct V3 { public var x, y, z: Double; public init(_ x: Double, _ y: Double, _ z: Double) { self.x = x; self.y = y; self.z = z } }
public struct Color { public var red, green, blue: Double; public init(_ r: Double, _ g: Double, _ b: Double) { red = r; green = g; blue = b } }
public struct MaterialValues { public var specularExponent, dissolution: Double; public var illumination: Int32; public var diffuse, ambient, specular: Color }
public struct LightValues { public var location: V3; public var specular, diffuse: Color }

public enum SceneObject {
    case sphere(center: V3, radius: Double, material: MaterialValues?)
    case boundingSphere(center: V3, radius: Double, material: MaterialValues?, children: [(V3, V3, V3, MaterialValues?)])
    case triangle(V3, V3, V3, MaterialValues?)
    case light(index: Int)
}

func tuple(_ v: V3) -> (Double, Double, Double) { (v.x, v.y, v.z) }
func tuple(_ c: Color) -> (Double, Double, Double) { (c.red, c.green, c.blue) }

/// A scene resident on one GPU (rtmi_scene_create / rtmi_scene_destroy).  Order of `objects` is
/// preserved: the first intersection in list order wins, as in RayTracingFunctions.swift:41-43.
public final class RtmiScene {
    var handle: OpaquePointer?

    public init(lights: [LightValues], objects: [SceneObject], device: Int32 = 0) {
        let cLights = lights.map { rtmi_light(loc: tuple($0.location), specular: tuple($0.specular), diffuse: tuple($0.diffuse)) }
        var cMaterials: [rtmi_material] = []
        var cTriangles: [rtmi_triangle] = []
        var cObjects: [rtmi_object] = []
        func materialIndex(_ m: MaterialValues?) -> Int32 {
            guard let m = m else { return -1 }  // nil stays nil
            cMaterials.append(rtmi_material(Ns: m.specularExponent, d: m.dissolution, illum: m.illumination, pad_: 0,
                                            Kd: tuple(m.diffuse), Ka: tuple(m.ambient), Ks: tuple(m.specular)))
            return Int32(cMaterials.count - 1)
        }
        func append(_ a: V3, _ b: V3, _ c: V3, _ m: MaterialValues?) {
            cTriangles.append(rtmi_triangle(v: (tuple(a), tuple(b), tuple(c)), material: materialIndex(m), pad_: 0))
        }
        for o in objects {
            switch o {
            case let .sphere(c, r, m):
                cObjects.append(rtmi_object(kind: Int32(RTMI_SPHERE.rawValue), material: materialIndex(m), center: tuple(c), radius: r, first: 0, count: 0))
            case let .boundingSphere(c, r, m, children):
                let first = Int64(cTriangles.count)
                for t in children { append(t.0, t.1, t.2, t.3) }
                cObjects.append(rtmi_object(kind: Int32(RTMI_BOUNDING_SPHERE.rawValue), material: materialIndex(m), center: tuple(c), radius: r,
                                            first: first, count: Int64(cTriangles.count) - first))
            case let .triangle(a, b, c, m):
                // consecutive triangles of the list form one run (their order is the order of cTriangles)
                if let last = cObjects.last, last.kind == Int32(RTMI_TRIANGLES.rawValue), last.first + last.count == Int64(cTriangles.count) {
                    cObjects[cObjects.count - 1].count += 1
                } else {
                    cObjects.append(rtmi_object(kind: Int32(RTMI_TRIANGLES.rawValue), material: -1, center: (0, 0, 0), radius: 0,
                                                first: Int64(cTriangles.count), count: 1))
                }
                append(a, b, c, m)
            case let .light(index):
                cObjects.append(rtmi_object(kind: Int32(RTMI_LIGHT.rawValue), material: -1, center: (0, 0, 0), radius: 0, first: Int64(index), count: 0))
            }
        }
        cLights.withUnsafeBufferPointer { lp in
            cMaterials.withUnsafeBufferPointer { mp in
                cObjects.withUnsafeBufferPointer { op in
                    cTriangles.withUnsafeBufferPointer { tp in
                        var desc = rtmi_scene_desc(lights: lp.baseAddress, n_lights: Int32(lp.count), n_materials: Int32(mp.count),
                                                   materials: mp.baseAddress, objects: op.baseAddress, triangles: tp.baseAddress,
                                                   n_objects: Int64(op.count), n_triangles: Int64(tp.count))
                        let rc = rtmi_scene_create(&desc, device, &handle)
                        precondition(rc == RTMI_OK.rawValue, String(cString: rtmi_last_error()))
                    }
                }
            }
        }
    }

    deinit { rtmi_scene_destroy(handle) }
}

/// The reference's globals (ViewController.swift:13-14).
public var imageWidth = 1000
public var imageHeight = 1000

/// raytraceWorld with the reference's own argument list (RayTracingFunctions.swift:145-155), so that the call at
/// ViewController.swift:160-179 stays as it is.  imageWidthPixels / imageHeightPixels and materialDictionary are
/// accepted and ignored, as in the reference (:170 uses the globals; the dictionary is never read).  The scene is
/// flattened and uploaded on every call; a caller that renders many frames of one scene keeps an RtmiScene and
/// calls the overload below.
public func raytraceWorld(camera: V3, cameraDirection: V3, focalLength: Double, imageWidthPixels: Int, imageHeightPixels: Int,
                          ambientLight: Color, lights: [LightValues], objects: [SceneObject],
                          materialDictionary: [String: MaterialValues], outputBitmap: inout [UInt8], pixelDone: (() -> Void)?) {
    let scene = RtmiScene(lights: lights, objects: objects)
    raytraceWorld(camera: camera, cameraDirection: cameraDirection, focalLength: focalLength, imageWidth: imageWidth,
                  imageHeight: imageHeight, ambientLight: ambientLight, scene: scene, outputBitmap: &outputBitmap, pixelDone: pixelDone)
}

/// What the library offers beyond the reference's behaviour; everything off = the reference.
public struct RenderExtensions {
    public var nearestHit = false           // RTMI_HIT_NEAREST: intersections.min(by: parameter) instead of intersections[0]
    public var shadowMaxDistance = false    // RTMI_FLAG_SHADOW_MAX_DISTANCE: only what lies before the light casts a shadow
    public var shadowsEnterBounded = false  // RTMI_FLAG_SHADOWS_ENTER_BOUNDED: the TODO at RayTracingFunctions.swift:240
    public var reflectionsEnterBounded = false  // RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED: mirrors show bounded meshes
    public init() {}
}

/// The same frame for a scene that is already resident on the GPU.
public func raytraceWorld(camera: V3, cameraDirection: V3, focalLength: Double, imageWidth: Int, imageHeight: Int,
                          pixelSubdivision: Double = 1, ambientLight: Color, scene: RtmiScene,
                          outputBitmap: inout [UInt8], pixelDone: (() -> Void)? = nil,
                          extensions: RenderExtensions = RenderExtensions()) {
    var cam = rtmi_camera(pos: tuple(camera), dir: tuple(cameraDirection), focal_length: focalLength)
    var p = rtmi_params()
    p.width = Int32(imageWidth)
    p.height = Int32(imageHeight)
    p.pixel_subdivision = pixelSubdivision  // :170 hard-codes 1; 0.5 = 4 samples per pixel, 0.25 = 16
    p.ambient = tuple(ambientLight)
    p.hit_policy = Int32((extensions.nearestHit ? RTMI_HIT_NEAREST : RTMI_HIT_LIST_ORDER).rawValue)
    // (the reference keeps no ray counts, so none are asked for: same pixels, and the library may skip rays that cannot change one)
    p.flags = RTMI_FLAG_NO_RAY_STATISTICS | (extensions.shadowMaxDistance ? RTMI_FLAG_SHADOW_MAX_DISTANCE : 0) |
              (extensions.shadowsEnterBounded ? RTMI_FLAG_SHADOWS_ENTER_BOUNDED : 0) |
              (extensions.reflectionsEnterBounded ? RTMI_FLAG_REFLECTIONS_ENTER_BOUNDED : 0)
    p.search = Int32(RTMI_SEARCH_CONE_TREE.rawValue)
    var counters = rtmi_counters()
    let rc = outputBitmap.withUnsafeMutableBufferPointer { rtmi_render(scene.handle, &cam, &p, $0.baseAddress, nil, &counters) }
    precondition(rc == RTMI_OK.rawValue, String(cString: rtmi_last_error()))
    let spp = Int64(rtmi_samples_per_pixel(pixelSubdivision))
    for _ in 0..<(counters.samples / spp) { pixelDone?() }  // once per rendered pixel (RayTracingFunctions.swift:112), after the frame
}
