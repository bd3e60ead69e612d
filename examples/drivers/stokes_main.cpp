// Stokes driver on the MI355X path, written against the FEDD:: operator surface the way the reference's driver is
// (feddlib/problems/tests/stokes/main.cpp:96-360, "unstructured" branch): same XML parameter files, same call sequence
//   Domain(comm, dim) x 2 -> MeshPartitioner::readAndPartition -> buildP2ofP1Domain -> BCBuilder::addBC ->
//   Stokes(...) -> addBoundaries -> initializeProblem -> assemble -> setBoundaries -> solve -> ExporterParaView.
// "Preconditioner Method" = "Monolithic" is the path built here (FROSch on the merged system, parametersPrec.xml);
// the Teko / Diagonal / Triangular block preconditioners of the reference are out of scope.
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>

#include "feddlib/core/FEDDCore.hpp"
#include "feddlib/core/FE/Domain.hpp"
#include "feddlib/core/General/BCBuilder.hpp"
#include "feddlib/core/General/ExporterParaView.hpp"
#include "feddlib/problems/specific/Stokes.hpp"

void zeroDirichlet2D(double* x, double* res, double t, const double* parameters) { res[0] = 0.; res[1] = 0.; }
void zeroDirichlet3D(double* x, double* res, double t, const double* parameters) { res[0] = 0.; res[1] = 0.; res[2] = 0.; }
void inflowParabolic2D(double* x, double* res, double t, const double* parameters) {
    double H = parameters[1];
    res[0] = 4 * parameters[0] * x[1] * (H - x[1]) / (H * H);
    res[1] = 0.;
}
void inflowParabolic3D(double* x, double* res, double t, const double* parameters) {
    double H = parameters[1];
    res[0] = 16 * parameters[0] * x[1] * (H - x[1]) * x[2] * (H - x[2]) / (H * H * H * H);
    res[1] = 0.;
    res[2] = 0.;
}

typedef default_sc SC;
typedef default_lo LO;
typedef default_go GO;
typedef default_no NO;

using namespace FEDD;

int main(int argc, char* argv[]) {
    typedef MeshPartitioner<SC, LO, GO, NO> MeshPartitioner_Type;
    typedef Teuchos::RCP<Domain<SC, LO, GO, NO> > DomainPtr_Type;
    std::string xmlProblemFile = "parametersProblem.xml", xmlPrecFile = "parametersPrec.xml", xmlSolverFile = "parametersSolver.xml";
    std::string outFile = "solutionStokes.txt";
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        auto val = [&](const char* key, std::string& dst) {
            const std::string k = std::string("--") + key + "=";
            if (a.compare(0, k.size(), k) == 0) { dst = a.substr(k.size()); return true; }
            return false;
        };
        if (val("problemfile", xmlProblemFile) || val("precfile", xmlPrecFile) || val("solverfile", xmlSolverFile) || val("out", outFile)) continue;
        std::cerr << "unknown option " << a << std::endl;
        return 2;
    }
    try {
        Teuchos::RCP<const Teuchos::Comm<int> > comm = Teuchos::rcp(new Teuchos::Comm<int>(0, 1));
        ParameterListPtr_Type parameterListProblem = Teuchos::getParametersFromXmlFile(xmlProblemFile);
        ParameterListPtr_Type parameterListPrec = Teuchos::getParametersFromXmlFile(xmlPrecFile);
        ParameterListPtr_Type parameterListSolver = Teuchos::getParametersFromXmlFile(xmlSolverFile);

        int dim = parameterListProblem->sublist("Parameter").get("Dimension", 3);
        std::string discVelocity = parameterListProblem->sublist("Parameter").get("Discretization Velocity", "P2");
        std::string discPressure = parameterListProblem->sublist("Parameter").get("Discretization Pressure", "P1");
        std::string meshType = parameterListProblem->sublist("Parameter").get("Mesh Type", "structured");
        int volumeID = parameterListProblem->sublist("Parameter").get("Volume ID", 0);
        std::string bcType = parameterListProblem->sublist("Parameter").get("BC Type", "parabolic");
        std::string precMethod = parameterListProblem->sublist("General").get("Preconditioner Method", "Monolithic");
        TEUCHOS_TEST_FOR_EXCEPTION(precMethod != "Monolithic", std::logic_error,
                                   "Preconditioner Method " << precMethod << ": only Monolithic is built (Teko and the block preconditioners are out of scope)");
        TEUCHOS_TEST_FOR_EXCEPTION(meshType != "unstructured", std::logic_error, "this driver reads unstructured meshes");
        TEUCHOS_TEST_FOR_EXCEPTION(discVelocity != "P2", std::logic_error, "this driver builds P2 / P1");

        ParameterListPtr_Type parameterListAll(new Teuchos::ParameterList(*parameterListProblem));
        parameterListAll->setParameters(*parameterListPrec);
        parameterListAll->setParameters(*parameterListSolver);

        Teuchos::RCP<Teuchos::Time> totalTime(Teuchos::TimeMonitor::getNewCounter("main: Total Time"));
        Teuchos::RCP<Teuchos::Time> buildMesh(Teuchos::TimeMonitor::getNewCounter("main: Build Mesh"));
        Teuchos::RCP<Teuchos::Time> solveTime(Teuchos::TimeMonitor::getNewCounter("main: Solve problem time"));
        DomainPtr_Type domainPressure, domainVelocity;
        int its = 0;
        {
            Teuchos::TimeMonitor totalTimeMonitor(*totalTime);
            {
                Teuchos::TimeMonitor buildMeshMonitor(*buildMesh);
                domainPressure.reset(new Domain<SC, LO, GO, NO>(comm, dim));
                domainVelocity.reset(new Domain<SC, LO, GO, NO>(comm, dim));
                MeshPartitioner_Type::DomainPtrArray_Type domainP1Array(1);
                domainP1Array[0] = domainPressure;
                ParameterListPtr_Type pListPartitioner = Teuchos::sublist(parameterListAll, "Mesh Partitioner");
                MeshPartitioner<SC, LO, GO, NO> partitionerP1(domainP1Array, pListPartitioner, "P1", dim);
                partitionerP1.readAndPartition(volumeID);
                domainVelocity->buildP2ofP1Domain(domainPressure);
            }
            std::vector<double> parameter_vec(1, parameterListProblem->sublist("Parameter").get("MaxVelocity", 1.));
            Teuchos::RCP<BCBuilder<SC, LO, GO, NO> > bcFactory(new BCBuilder<SC, LO, GO, NO>());
            if (!bcType.compare("parabolic")) parameter_vec.push_back(1.);
            else if (!bcType.compare("parabolic_benchmark")) parameter_vec.push_back(.41);
            else TEUCHOS_TEST_FOR_EXCEPTION(true, std::logic_error, "Select a valid boundary condition.");
            if (dim == 2) {
                bcFactory->addBC(zeroDirichlet2D, 1, 0, domainVelocity, "Dirichlet", dim);
                bcFactory->addBC(inflowParabolic2D, 2, 0, domainVelocity, "Dirichlet", dim, parameter_vec);
            } else {
                bcFactory->addBC(zeroDirichlet3D, 1, 0, domainVelocity, "Dirichlet", dim);
                bcFactory->addBC(inflowParabolic3D, 2, 0, domainVelocity, "Dirichlet", dim, parameter_vec);
            }
            if (!bcType.compare("parabolic_benchmark"))
                bcFactory->addBC(dim == 2 ? zeroDirichlet2D : zeroDirichlet3D, 4, 0, domainVelocity, "Dirichlet", dim);

            Stokes<SC, LO, GO, NO> stokes(domainVelocity, discVelocity, domainPressure, discPressure, parameterListAll);
            domainVelocity->info();
            domainPressure->info();
            stokes.info();
            {
                Teuchos::TimeMonitor solveTimeMonitor(*solveTime);
                stokes.addBoundaries(bcFactory);
                stokes.initializeProblem();
                stokes.assemble();
                stokes.setBoundaries();
                its = stokes.solve();
            }
            std::cout << "iterations " << its << " relres " << stokes.getLastRelativeResidual() << std::endl;
            Teuchos::RCP<const MultiVector<SC, LO, GO, NO> > exportSolutionV = stokes.getSolution()->getBlock(0);
            Teuchos::RCP<const MultiVector<SC, LO, GO, NO> > exportSolutionP = stokes.getSolution()->getBlock(1);
            std::ofstream out(outFile);
            out << std::setprecision(17);
            for (size_t i = 0; i < exportSolutionV->getLocalLength(); ++i) out << i << " " << exportSolutionV->getData(0)[i] << "\n";
            const size_t off = exportSolutionV->getLocalLength();
            for (size_t i = 0; i < exportSolutionP->getLocalLength(); ++i) out << off + i << " " << exportSolutionP->getData(0)[i] << "\n";
            if (parameterListAll->sublist("General").get("ParaViewExport", false)) {
                Teuchos::RCP<ExporterParaView<SC, LO, GO, NO> > exParaVelocity(new ExporterParaView<SC, LO, GO, NO>());
                Teuchos::RCP<ExporterParaView<SC, LO, GO, NO> > exParaPressure(new ExporterParaView<SC, LO, GO, NO>());
                exParaVelocity->setup("velocity", domainVelocity->getMesh(), domainVelocity->getFEType());
                exParaVelocity->addVariable(exportSolutionV, "u", "Vector", dim, domainVelocity->getMapUnique());
                exParaPressure->setup("pressure", domainPressure->getMesh(), domainPressure->getFEType());
                exParaPressure->addVariable(exportSolutionP, "p", "Scalar", 1, domainPressure->getMapUnique());
                exParaVelocity->save(0.0);
                exParaPressure->save(0.0);
            }
        }
        Teuchos::TimeMonitor::report(std::cout);
    } catch (const std::exception& e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
